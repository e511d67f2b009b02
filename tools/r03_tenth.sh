cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03j; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "line" > $o/tests_line.log 2>&1; echo "rc=$?" >> $o/tests_line.log
python3 tools/ablate.py run mdpp_continuous_line.hip line numpy shipped > $o/line.txt 2>&1
python3 tools/ablate.py run mdpp_continuous_line.hip line philox shipped >> $o/line.txt 2>&1
bash tools/pmc_sq.sh line 512 2 > $o/sq_line.txt 2>&1
tail -15 $o/tests_line.log; cat $o/line.txt; cat $o/sq_line.txt
