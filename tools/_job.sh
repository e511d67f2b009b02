cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
{
for i in 1 2 3; do timeout 900 python3 tools/ablate.py run mdpp_image.hip cfg4 numpy base res0 res16 res32 res64; done
} > $o/ablate_res.txt 2>&1
cut -c1-200 $o/ablate_res.txt | grep -v "^$" | tail -16
