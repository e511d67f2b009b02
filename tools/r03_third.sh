# round 3, third GPU call: the whole -m gpu suite, lean ablations (round 2), cfg5 Philox after the packed Box-Muller,
# kernel timeline of the bench's collective leg
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03c; mkdir -p $o
timeout 1500 python3 -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $o/tests_gpu.log
python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 numpy shipped ah2 ah2ld2 ah2ld3 ah2d16 ah2st3 > $o/ablate_lean2.txt 2>&1
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox shipped > $o/cfg5_philox.txt 2>&1
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 numpy shipped >> $o/cfg5_philox.txt 2>&1
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg3 numpy shipped >> $o/cfg5_philox.txt 2>&1
python3 tools/ablate.py run mdpp_discrete_quiet.hip cfg2_noise philox shipped >> $o/cfg5_philox.txt 2>&1
rm -rf gpurun_out/trace_coll
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_coll -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-workloads --no-single-step > $o/bench_traced.json 2> $o/bench_traced.err
python3 - <<'PY' > gpurun_out/r03c/timeline.txt
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trace_coll/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if "rollout" in r["Kernel_Name"] or "ccl" in r["Kernel_Name"].lower() or "Copy" in r["Kernel_Name"] or "copy" in r["Kernel_Name"]]
print("n kernels", len(rows), "selected", len(sel))
prev_end = None
for r in sel[-140:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%12.1f us  dur %8.1f us  q%s  %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:70]))
PY
rm -rf gpurun_out/trace_coll
tail -3 $o/tests_gpu.log; cat $o/ablate_lean2.txt $o/cfg5_philox.txt; tail -50 $o/timeline.txt
