cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "specialised_kernels_equal and (rdist or s50)" 2>&1 | grep -v "^$" | tail -4
