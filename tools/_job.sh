cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
{
for i in 1 2 3; do timeout 600 python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 numpy base r1d32 r1d40; done
} > $o/ablate_d.txt 2>&1
cut -c1-10,85-200 $o/ablate_d.txt | grep -v "^$" | tail -14
