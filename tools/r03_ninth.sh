cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03i; mkdir -p $o
timeout 1800 python3 -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $o/tests_gpu.log
tail -30 $o/tests_gpu.log
