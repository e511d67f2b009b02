cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q -k "many_boxes or line_reward_checkpoint or sparse_term" 2>&1 | tail -5
