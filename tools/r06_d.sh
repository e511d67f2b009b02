cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
python3 tools/pmc_workloads.py 3 cfg2:numpy:65536:512 cfg2:philox:65536:512 cfg3:numpy:65536:512 cfg4:numpy:8192:512 cfg5:numpy:65536:512 cfg5:philox:65536:512 cfg2_noise:numpy:65536:512 cfg2_noise:philox:65536:512 d_s50_delay4:numpy:65536:512 d_s24_rdist:numpy:65536:512 cfg2_per_env:numpy:65536:512 img100_all:numpy:8192:64 d_s8_rn0:numpy:65536:512 d_s8_rn0:philox:65536:512 d_s50_rn0:numpy:65536:512 c_d2_n0:numpy:65536:512 c_d2_n0:philox:65536:512 2>&1 | grep -o "workload=[a-z0-9_]* rng=[a-z]*\|make_env_s.*"  | paste - -
echo "wall $(( $(date +%s) - t0 )) s"
