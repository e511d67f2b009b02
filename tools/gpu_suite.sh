cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/gpu_suite; mkdir -p $o
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $o/pytest_gpu.txt
cat $o/pytest_gpu.txt
