cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -q -x -k "graph or step1 or image" 2>&1 | tail -8
