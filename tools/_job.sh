cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sweep.py -m gpu -q 2>&1 | tail -12 | cut -c1-400
