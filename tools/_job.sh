cd $GRAFT_REPO_ROOT
python3 tools/ablate.py run mdpp_continuous_fast.hip c_d2_n0 numpy shipped nowait nocons consonly 2>&1 | tail -4
