cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg5 or continuous or cfg3 or line or default_target" > $o/tests_c5.log 2>&1; tail -3 $o/tests_c5.log
{
for i in 1 2; do timeout 600 python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 numpy base old; done
for i in 1 2; do timeout 600 python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox base old; done
for i in 1 2; do timeout 600 python3 tools/ablate.py run mdpp_continuous_fast.hip cfg3 numpy base old; done
} > $o/ablate_c5.txt 2>&1
cut -c1-10,75-200 $o/ablate_c5.txt | grep -v "^$" | tail -14
