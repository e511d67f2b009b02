cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised_kernels_equal and per_env" 2>&1 | tail -3
