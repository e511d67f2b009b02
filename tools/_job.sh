cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/step1
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/step1/tests_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -15 gpurun_out/step1/tests_gpu.log
