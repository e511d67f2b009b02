# round 6, second lease: parity of this round's kernels (sigma-0 draws, the wide line fit, step1w blobs), then the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06b; mkdir -p $o
python3 -m pytest tests/test_gpu_sweep.py -m gpu -q -x -k "sigma_zero or wide_one_step" 2>&1 | tail -8
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sigma_zero or beyond_eight or line_reward or c_line or stepwise" 2>&1 | tail -8
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_stdout.txt 2> $o/bench_stderr.txt
echo "rc $? wall $(( $(date +%s) - t0 )) s"
cp gpurun_out/bench_detail.json $o/bench_detail.json
tail -n 1 $o/bench_stdout.txt
tail -5 $o/bench_stderr.txt
