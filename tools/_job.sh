cd $GRAFT_REPO_ROOT
bash tools/validate_all.sh
