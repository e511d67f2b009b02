# round 3, second GPU call: RCCL / boundary tests, the Philox start-state stream against the oracle, lean-kernel ablations with honest reads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03b; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_boundary.py -m gpu -x -q > $o/tests_boundary.log 2>&1; echo "boundary rc=$?" >> $o/tests_boundary.log
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "philox or sharding or lean" > $o/tests_philox.log 2>&1; echo "philox rc=$?" >> $o/tests_philox.log
python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 numpy shipped ld1 ld2 ld3 ah8 ah2 noload nostore > $o/ablate_lean.txt 2>&1
python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 philox shipped > $o/ablate_lean_philox.txt 2>&1
tail -4 $o/tests_boundary.log; tail -6 $o/tests_philox.log; cat $o/ablate_lean.txt $o/ablate_lean_philox.txt
