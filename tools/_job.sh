cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bench_size" 2>&1 | tail -3
