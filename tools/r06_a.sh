# round 6, first lease: the contract line (last stdout line, <= 4 KiB), the phases' wall time, the tests this round touched
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06a; mkdir -p $o
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_stdout.txt 2> $o/bench_stderr.txt
echo "rc $? wall $(echo "$(date +%s.%N) - $t0" | bc)" | tee -a $o/bench_stderr.txt
cp gpurun_out/bench_detail.json $o/bench_detail.json
tail -c 6000 $o/bench_stdout.txt
python3 -m pytest tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -5
