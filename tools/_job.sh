cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden and (i_ or image)" 2>&1 | tail -5
