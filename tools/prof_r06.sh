# usage: bash tools/prof_r06.sh   (GPU box) -- the round-6 records under profiles/ (copied there from gpurun_out/r06p):
#   the driver's bench command (stdout as the driver sees it + the detail record + wall time); rocprofv3 kernel stats of the same command
#   and of each leg of the headline kernel on its own; the dispatch table; the bench line under torch.distributed.run (one RCCL rank)
#   and at world 8 on this one GPU (gloo).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06p; mkdir -p $o
t0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_driver_stdout.txt 2>/dev/null
echo "driver command: rc $? wall $(( $(date +%s) - t0 )) s, last stdout line $(tail -n 1 $o/r06_bench_driver_stdout.txt | wc -c) bytes" | tee $o/r06_bench_driver_wall.txt
tail -n 1 $o/r06_bench_driver_stdout.txt > $o/r06_bench_driver_argv.json
cp gpurun_out/bench_detail.json $o/r06_bench_detail.json
for leg in rotating replayed; do
  rm -rf gpurun_out/prof_r06
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --only-leg $leg --detail-out $o/x.json > $o/bench_under_rocprof_$leg.log 2>&1
  grep '^{"metric"' $o/bench_under_rocprof_$leg.log > $o/r06_bench_under_rocprof_$leg.json
  find gpurun_out/prof_r06 -name "*kernel_stats.csv" -exec cp {} $o/r06_kernel_stats_$leg.csv \;
done
rm -rf gpurun_out/prof_r06
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --detail-out $o/r06_bench_under_rocprof_detail.json > $o/bench_under_rocprof.log 2>&1
grep '^{"metric"' $o/bench_under_rocprof.log > $o/r06_bench_under_rocprof.json
find gpurun_out/prof_r06 -name "*kernel_stats.csv" -exec cp {} $o/r06_kernel_stats.csv \;
rm -rf gpurun_out/prof_r06 $o/x.json
python3 tools/dispatch_table.py $o/dispatch.md 2>&1 | tail -1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --detail-out $o/x.json 2>/dev/null | tail -n 1 > $o/r06_bench_torchrun1.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 8 --steps 10 --warmup 3 --envs 8192 --backend gloo --no-single-step --detail-out $o/r06_bench_world8_gloo_detail.json 2>/dev/null | tail -n 1 > $o/r06_bench_world8_gloo_one_gpu.json
rm -f $o/x.json
python3 - <<'PY'
import json, csv
o = "gpurun_out/r06p/"
d = json.loads(open(o + "r06_bench_driver_argv.json").read())
print(json.dumps(d)[:1600])
print("timing", json.load(open(o + "r06_bench_detail.json"))["timing_s"])
for leg in ("rotating", "replayed"):
    for row in csv.DictReader(open(o + f"r06_kernel_stats_{leg}.csv")):
        if "mdpp::" in row["Name"] and float(row["Percentage"]) > 0.5:
            print(leg, "   %-100s calls %5s avg %10.1f us" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3))
w8 = json.loads(open(o + "r06_bench_world8_gloo_one_gpu.json").read())
print("world 8 (gloo, one GPU): n_gpus", w8["n_gpus"], "value", w8["value"], "bytes", len(json.dumps(w8)))
PY
