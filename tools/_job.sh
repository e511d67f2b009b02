cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_gpu_sweep.py -m gpu -q 2>&1 | tail -12 | cut -c1-600
timeout 2400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "continuous" 2>&1 | tail -3 | cut -c1-600
