cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q -k "_x" 2>&1 | tail -40 | cut -c1-300
