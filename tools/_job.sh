cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step1 or stepwise or full_size" 2>&1 | tail -4
