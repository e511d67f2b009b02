# usage: bash tools/pmc_traffic.sh <workload> <fuse> [launches] [rng=philox] [disable=...]  (GPU box)
# HBM traffic of the fused rollouts of one workload: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (MI355X_MICROARCH.md: they do not fit one pass), summed over every mdpp::
# kernel, per env step.  Writes gpurun_out/traffic_<workload>.json.
w=$1; F=$2; L=${3:-4}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${w}_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${w}_$c -- python3 tools/run_variant.py - $L $w $F ${@:4} > gpurun_out/pmc_${w}_$c.log 2>&1
done
python3 - "$w" "$F" "$L" <<'PY'
import csv, glob, json, sys
w, F, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, ".")
import bench
N = bench.WORKLOADS[w]["envs"]
tot = {}
kern = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True)
    s = 0.0
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "mdpp::" in r["Kernel_Name"]:
                s += float(r["Counter_Value"])
                k = r["Kernel_Name"].split("(")[0]
                kern.setdefault(k, {}).setdefault(c, 0.0)
                kern[k][c] += float(r["Counter_Value"])
    tot[c] = s
steps = N * F * L
# reset kernels of the constructor are included (negligible next to L launches of F steps)
read_b = 2 * tot["FETCH_SIZE"] * 1024          # gfx950: 128-B read requests tallied at 64 B
write_b = tot["WRITE_SIZE"] * 1024
alg = bench.WORKLOADS[w]["alg_bytes_fused"]
out = {"what": f"HBM traffic of the fused rollouts of {w} ({N} envs, {F} steps/launch, {L} launches), rocprofv3 --pmc, "
               "separate passes, summed over all mdpp:: kernels",
       "command": f"bash tools/pmc_traffic.sh {w} {F} {L}",
       "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
       "correction": "gfx950: FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section) -> "
                     "read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-B-per-lane stores, "
                     "uncalibrated for narrower ones",
       "read_bytes": read_b, "write_bytes": write_b, "env_steps": steps,
       "traffic_bytes_per_env_step": (read_b + write_b) / steps, "algorithmic_bytes_per_env_step": alg,
       "envs": N, "fuse": F, "per_kernel_KB": kern}
json.dump(out, open(f"gpurun_out/traffic_{w}.json", "w"), indent=1)
print(w, "traffic B/env-step", out["traffic_bytes_per_env_step"], "algorithmic", alg)
PY
rm -rf gpurun_out/pmc_${w}_FETCH_SIZE gpurun_out/pmc_${w}_WRITE_SIZE
