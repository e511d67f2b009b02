# usage: bash tools/pmc_step1.sh <workload> [n] [rng] [opts]   (GPU box)  -- SQ counters of the one-step kernels, per wave and launch
w=$1; n=${2:-100}; rng=${3:-numpy}; opts=${4:-}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
sets=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH")
for c in "${sets[@]}"; do
  t=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/sq1_$t
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/sq1_$t -- python3 tools/step1_loop.py $w $n $rng $opts > gpurun_out/sq1_$t.log 2>&1
done
python3 - "$n" <<'PY'
import csv, glob, sys
n = int(sys.argv[1])
tot = {}
for f in glob.glob("gpurun_out/sq1_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "mdpp::" not in k or "reset" in k:
            continue
        tot.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    waves = d.get("SQ_WAVES", 0) or 1
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:22s} {v:16.0f}   per launch {v / n:14.0f}   per wave {v / waves:10.1f}")
PY
rm -rf gpurun_out/sq1_SQ_*
