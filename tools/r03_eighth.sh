cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03h; mkdir -p $o
timeout 1800 python3 -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $o/tests_gpu.log
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox shipped > $o/timing.txt 2>&1
python3 tools/ablate.py run mdpp_discrete_quiet.hip cfg2_noise philox shipped >> $o/timing.txt 2>&1
python3 tools/ablate.py run mdpp_discrete_quiet.hip cfg2_noise numpy shipped >> $o/timing.txt 2>&1
python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 philox shipped >> $o/timing.txt 2>&1
tail -3 $o/tests_gpu.log; cat $o/timing.txt
