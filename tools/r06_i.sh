cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in cfg3 cfg5 c_d2_n0; do python3 tools/ablate.py run mdpp_continuous_fast.hip $w numpy c0 c1 c0 c1 2>&1 | grep -v amdgpu.ids | sed "s/^/$w /" | cut -c1-12,105-; done
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox c0 c1 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg5ph /" | cut -c1-12,105-
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg3 numpy c0 c1 transition_noise=None 2>&1 | tail -2 | cut -c1-12,100-
