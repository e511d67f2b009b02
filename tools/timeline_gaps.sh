# usage: bash tools/timeline_gaps.sh   (GPU box) -- rocprofv3 kernel + memory-copy trace of a short bench run: duration of every
# k_discrete_rollout_lean launch, the gap to the next one and what ran in between (the legs: none, replay, last_row, full)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --gpus 1 --steps ${TL_STEPS:-12} --warmup 3 --no-cpu-baseline --no-pmc --no-workloads ${TL_ARGS} > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
for f in glob.glob("gpurun_out/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", ""), ""))
ev.sort()
import os
pat = os.environ.get("TL_KERNEL", "rollout_lean")
lean = [k for k, e in enumerate(ev) if pat in e[2]]
print("lean kernels:", len(lean))
# gaps between consecutive lean kernels, grouped
gaps = []
for a, b in zip(lean[:-1], lean[1:]):
    between = [ev[k][2][:28] + "(%.1f)" % ((ev[k][1] - ev[k][0]) / 1e3) for k in range(a + 1, b)]
    gaps.append(((ev[b][0] - ev[a][1]) / 1e3, (ev[a][1] - ev[a][0]) / 1e3, between))
for g in gaps:
    print("dur %7.1f  gap-to-next %8.1f  between: %s" % (g[1], g[0], " ".join(g[2][:4])))
PY
rm -rf gpurun_out/tl
