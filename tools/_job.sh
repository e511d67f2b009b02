cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q -k "step1 or full_size or graph or continuous or cfg3 or checkpoint or c_" 2>&1 | tail -4
python3 tools/bench_step1.py cfg3 --check --philox 2>&1 | grep "timing\|mismatch" | cut -c1-40,100-330
