cd $GRAFT_REPO_ROOT
python3 tools/probe_pictures.py 2>&1 | tail -8
