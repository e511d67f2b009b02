cd $GRAFT_REPO_ROOT
python3 tools/ablate_step1.py run mdpp_continuous_step1.hip cfg3 numpy wide narrow wide 2>&1 | cut -c1-200
python3 tools/ablate_step1.py run mdpp_continuous_step1.hip cfg5 philox wide narrow 2>&1 | cut -c1-200
python3 tools/ablate_step1.py run mdpp_continuous_step1.hip cfg5 numpy wide narrow 2>&1 | cut -c1-200
