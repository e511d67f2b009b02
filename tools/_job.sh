cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -4
