# usage: bash tools/prof_r03.sh   (GPU box) -- the round-3 records under profiles/:
#   the driver's bench command (live PMC traffic for every workload leg), rocprofv3 kernel stats of the same command,
#   offline traffic records per workload, the bench line under torch.distributed.run with one rank.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03p; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > $o/r03_bench_driver_argv.json
rm -rf gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_under_rocprof.log 2>&1
grep '^{"metric"' $o/bench_under_rocprof.log > $o/r03_bench_under_rocprof.json
find gpurun_out/prof_r03 -name "*kernel_stats.csv" -exec cp {} $o/r03_kernel_stats.csv \;
rm -rf gpurun_out/prof_r03
for w in cfg2 cfg3 cfg4 cfg5; do
  bash tools/pmc_traffic.sh $w 512 3 > /dev/null 2>&1; cp gpurun_out/traffic_$w.json $o/r03_traffic_$w.json
done
for w in cfg2 cfg5 cfg2_noise; do
  bash tools/pmc_traffic.sh $w 512 3 rng=philox > /dev/null 2>&1; cp gpurun_out/traffic_$w.json $o/r03_traffic_${w}_philox.json
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > $o/r03_bench_torchrun1.json
rm -rf gpurun_out/pmc_*
python3 - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/r03p/r03_bench_driver_argv.json").read())
print("value", d["value"], "none", d["value_none"], "frac", d["roofline"]["frac"], "replayed", d["roofline"]["frac_replayed"], "launch_us", d["roofline"]["launch_us"], "traffic", d["roofline"]["traffic"])
for k, v in (d["workloads"] or {}).items():
    print(k, {kk: v.get(kk) for kk in ("env_steps_per_s", "launch_us", "frac", "traffic", "error")})
u = json.loads(open("gpurun_out/r03p/r03_bench_under_rocprof.json").read())
print("under rocprof: launch_us", u["roofline"]["launch_us"], {k: v.get("launch_us") for k, v in (u["workloads"] or {}).items()})
for r in csv.DictReader(open("gpurun_out/r03p/r03_kernel_stats.csv")):
    if "mdpp::" in r["Name"] and float(r["Percentage"]) > 0.5:
        print("%-110s calls %5s avg %10.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
t = json.loads(open("gpurun_out/r03p/r03_bench_torchrun1.json").read())
print("torchrun1: value", t["value"], "none", t["value_none"], t["config"]["collective"])
PY
