cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "continuous or cfg3 or cfg5 or line or default_target" > $o/tests_c.log 2>&1; tail -3 $o/tests_c.log
{
for i in 1 2 3; do timeout 600 python3 tools/ablate.py run mdpp_continuous_fast.hip cfg3 numpy base r0; done
} > $o/ablate_c3d.txt 2>&1
cut -c1-10,60-200 $o/ablate_c3d.txt | grep -v "^$" | tail -40
