# usage: bash tools/prof_r04.sh   (GPU box) -- the round-4 records under profiles/:
#   the driver's bench command; rocprofv3 kernel stats of the same command AND of each leg of the headline kernel on its own
#   (`--only-leg rotating` / `replayed`: one leg per csv); offline traffic records of the kernels that
#   changed this round; the bench line under torch.distributed.run with one rank.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r04p; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > $o/r04_bench_driver_argv.json
for leg in rotating replayed; do
  rm -rf gpurun_out/prof_r04
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --only-leg $leg > $o/bench_under_rocprof_$leg.log 2>&1
  grep '^{"metric"' $o/bench_under_rocprof_$leg.log > $o/r04_bench_under_rocprof_$leg.json
  find gpurun_out/prof_r04 -name "*kernel_stats.csv" -exec cp {} $o/r04_kernel_stats_$leg.csv \;
done
rm -rf gpurun_out/prof_r04
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_under_rocprof.log 2>&1
grep '^{"metric"' $o/bench_under_rocprof.log > $o/r04_bench_under_rocprof.json
find gpurun_out/prof_r04 -name "*kernel_stats.csv" -exec cp {} $o/r04_kernel_stats.csv \;
rm -rf gpurun_out/prof_r04
for w in cfg5 cfg2_noise; do
  bash tools/pmc_traffic.sh $w 512 3 > /dev/null 2>&1; cp gpurun_out/traffic_$w.json $o/r04_traffic_$w.json
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > $o/r04_bench_torchrun1.json
rm -rf gpurun_out/pmc_*
python3 - <<'PY'
import json, csv
o = "gpurun_out/r04p/"
d = json.loads(open(o + "r04_bench_driver_argv.json").read())
r = d["roofline"]
print("value", d["value"], "runs", d["value_runs"], "none", d["value_none"], "frac", r["frac"], "replayed", r.get("frac_replayed"),
      "launch_us", r["launch_us"], r["launch_us_runs"], "traffic", r["traffic"], "peaks", r.get("peak_measured"))
for k, v in (d["workloads"] or {}).items():
    print(k, {kk: v.get(kk) for kk in ("env_steps_per_s", "launch_us", "launch_us_runs", "frac", "traffic", "error")})
for leg in ("rotating", "replayed"):
    u = json.loads(open(o + f"r04_bench_under_rocprof_{leg}.json").read())
    print(leg, "under rocprof: launch_us", u["roofline"]["launch_us"], u["roofline"]["launch_us_runs"])
    for row in csv.DictReader(open(o + f"r04_kernel_stats_{leg}.csv")):
        if "mdpp::" in row["Name"] and float(row["Percentage"]) > 0.5:
            print("   %-100s calls %5s avg %10.1f us" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3))
for row in csv.DictReader(open(o + "r04_kernel_stats.csv")):
    if "mdpp::" in row["Name"] and float(row["Percentage"]) > 0.5:
        print("%-110s calls %5s avg %10.1f us" % (row["Name"][:110], row["Calls"], float(row["AverageNs"]) / 1e3))
t = json.loads(open(o + "r04_bench_torchrun1.json").read())
print("torchrun1: value", t["value"], "none", t["value_none"], t["config"]["collective"], t["multi_rank_diagnostics"] and {k: v for k, v in t["multi_rank_diagnostics"].items() if k not in ("per_rank", "note")})
PY
