# usage: bash tools/pmc_valu_ab.sh <workload> [<workload> ...]   (GPU box)
# vector instructions per fused launch of a workload's rollout kernel on the default dispatch and with MDPP_OPT_NO_SIGMA0
# (rocprofv3 --pmc SQ_INSTS_VALU, a pass of its own per run; bench.py --only-leg starts no child profiler).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/valu_ab; mkdir -p $o
for w in "$@"; do
  for tag in default NO_SIGMA0; do
    dis=""; [ $tag = NO_SIGMA0 ] && dis="--disable NO_SIGMA0"
    rm -rf /tmp/pmc_ab
    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pmc_ab -- python3 bench.py --workload $w $dis --only-leg rotating --steps 3 --warmup 1 --repeats 1 --detail-out $o/x.json > /dev/null 2>&1
    python3 - "$w" "$tag" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/pmc_ab/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == "SQ_INSTS_VALU" and "rollout" in r["Kernel_Name"]]
by = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:90]
    by.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in by.items():
    print(sys.argv[1], sys.argv[2], k, "launches", len(v), "SQ_INSTS_VALU per launch %.0f" % (sum(v) / len(v)), "per SIMD %.0f" % (sum(v) / len(v) / 1024))
PY
  done
done
