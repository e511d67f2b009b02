cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
{
for i in 1 2; do timeout 600 python3 tools/ablate.py run mdpp_discrete_lean_npnoise.hip cfg2_noise numpy base ah2; done
for i in 1 2; do timeout 600 python3 tools/ablate.py run mdpp_discrete_lean_noise.hip cfg2_noise philox base ah2; done
} > $o/ablate_n.txt 2>&1
cut -c1-10,85-200 $o/ablate_n.txt | grep -v "^$" | tail -12
