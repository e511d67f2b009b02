# usage: bash tools/pmc_sq.sh <workload> <fuse> [launches]   [rng=philox] [disable=NO_CFAST,...]   (GPU box)
# SQ instruction-mix counters of the workload's rollout kernels, per wave and env step.
w=$1; F=$2; L=${3:-2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# (SQ_MORE=1: where the issue cycles go -- active cycles per unit, integer / float64 / transcendental instruction counts)
sets=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH")
if [ -n "$SQ_MORE" ]; then
  sets+=("SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F32")
fi
for c in "${sets[@]}"; do
  t=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/sq_$t
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/sq_$t -- python3 tools/run_variant.py - $L $w $F ${@:4} > gpurun_out/sq_$t.log 2>&1
done
python3 - "$w" "$F" "$L" <<'PY'
import csv, glob, sys
w, F, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tot = {}
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
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
        print(f"   {c:22s} {v:16.0f}   per wave-step {v / waves / F:10.1f}")
PY
rm -rf gpurun_out/sq_SQ_*
