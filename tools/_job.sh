bash tools/validate_all.sh
