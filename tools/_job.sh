cd $GRAFT_REPO_ROOT
python3 tools/bench_step1.py cfg5 --check 2>&1 | grep "mismatch\|timing" | cut -c1-330
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step1 or full_size" 2>&1 | tail -3
