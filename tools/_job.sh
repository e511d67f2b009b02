cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "noise" > $o/tests_n.log 2>&1; tail -3 $o/tests_n.log
timeout 600 python3 tools/soak_noise.py numpy > $o/soak_noise.log 2>&1; tail -2 $o/soak_noise.log
{
for i in 1 2 3; do timeout 600 python3 tools/ablate.py run mdpp_discrete_lean_npnoise.hip cfg2_noise numpy base old; done
} > $o/ablate_n2.txt 2>&1
cut -c1-10,85-200 $o/ablate_n2.txt | grep -v "^$" | tail -8
