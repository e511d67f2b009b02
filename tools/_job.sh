cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "c_x or step1 or graph" 2>&1 | tail -3
