# usage: bash tools/prof_r05.sh   (GPU box) -- the round-5 records under profiles/ (copied there from gpurun_out/r05p):
#   the driver's bench command; rocprofv3 kernel stats of the same command and of each leg of the headline kernel on its own;
#   rocprofv3 kernel stats of the ONE-STEP kernels (tools/step1_loop.py: mdpp_step launches only); their timing beside the
#   rollout kernels with K = 1 (tools/bench_step1.py); the bench line at world 8 on this one GPU (gloo) and under
#   torch.distributed.run with one RCCL rank.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r05p; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > $o/r05_bench_driver_argv.json
for leg in rotating replayed; do
  rm -rf gpurun_out/prof_r05
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --only-leg $leg > $o/bench_under_rocprof_$leg.log 2>&1
  grep '^{"metric"' $o/bench_under_rocprof_$leg.log > $o/r05_bench_under_rocprof_$leg.json
  find gpurun_out/prof_r05 -name "*kernel_stats.csv" -exec cp {} $o/r05_kernel_stats_$leg.csv \;
done
rm -rf gpurun_out/prof_r05
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_under_rocprof.log 2>&1
grep '^{"metric"' $o/bench_under_rocprof.log > $o/r05_bench_under_rocprof.json
find gpurun_out/prof_r05 -name "*kernel_stats.csv" -exec cp {} $o/r05_kernel_stats.csv \;
# the one-step kernels alone: 2000 mdpp_step launches per workload and RNG
: > $o/r05_step1_kernel_stats.csv
for spec in "cfg2 numpy" "cfg2 philox" "cfg3 numpy" "cfg5 numpy" "cfg5 philox"; do
  set -- $spec
  rm -rf gpurun_out/prof_r05
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -- python3 tools/step1_loop.py $1 2000 $2 > /dev/null 2>&1
  find gpurun_out/prof_r05 -name "*kernel_stats.csv" -exec grep -h "step1\|rollout_fast" {} \; | sed "s/^/$1,$2,/" >> $o/r05_step1_kernel_stats.csv
done
rm -rf gpurun_out/prof_r05
python3 tools/bench_step1.py cfg2 cfg3 cfg5 --philox 2>/dev/null | grep timing > $o/r05_step1_timing.jsonl
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > $o/r05_bench_torchrun1.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 8 --steps 10 --warmup 3 --envs 8192 --backend gloo --peer-copy --no-single-step 2>/dev/null | grep '^{"metric"' > $o/r05_bench_world8_gloo_one_gpu.json
python3 - <<'PY'
import json, csv
o = "gpurun_out/r05p/"
d = json.loads(open(o + "r05_bench_driver_argv.json").read())
r = d["roofline"]
print("value", d["value"], "none", d["value_none"], "frac", r["frac"], "replayed", r.get("frac_replayed"), "frac_value", r.get("frac_value"),
      "launch_us", r["launch_us"], "traffic", r["traffic"], "valu_frac", r.get("valu_frac"), "peaks", r.get("peak_measured"))
print("single", json.dumps(d["single_step"])[:900])
for k, v in (d["workloads"] or {}).items():
    print(k, {kk: v.get(kk) for kk in ("launch_us", "frac", "valu_frac", "bound", "traffic", "error")}, json.dumps((v.get("single_step") or {}))[:400])
for leg in ("rotating", "replayed"):
    for row in csv.DictReader(open(o + f"r05_kernel_stats_{leg}.csv")):
        if "mdpp::" in row["Name"] and float(row["Percentage"]) > 0.5:
            print(leg, "   %-100s calls %5s avg %10.1f us" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3))
print(open(o + "r05_step1_kernel_stats.csv").read())
w8 = json.loads(open(o + "r05_bench_world8_gloo_one_gpu.json").read())
print("world 8 (gloo, one GPU): n_gpus", w8["n_gpus"], "value", w8["value"], "ranks", len(w8["multi_rank_diagnostics"]["per_rank"]), "peer_copy", {k: w8["collective_legs"]["peer_copy"].get(k) for k in ("timeouts", "checked_against_rccl", "error")})
PY
