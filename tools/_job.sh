cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py -m gpu -q -x -k "c_x or continuous or cfg3 or cfg5" 2>&1 | tail -3
python3 tools/ablate.py run mdpp_continuous_fast.hip c_d2_n0 numpy transition_noise=None reward_noise=None shipped 2>&1 | tail -1
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg3 numpy shipped 2>&1 | tail -1
