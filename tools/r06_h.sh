cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/ablate.py run mdpp_discrete_quiet_nu.hip d_s24_rdist numpy norows rows norows rows 2>&1 | grep -v amdgpu.ids | sed "s/^/d_s24_rdist /" | cut -c1-30,118-
for w in d_s50_rn0 d_s50_delay4; do python3 tools/ablate.py run mdpp_discrete_quiet.hip $w numpy norows rows norows rows 2>&1 | grep -v amdgpu.ids | sed "s/^/$w /" | cut -c1-30,110-; done
