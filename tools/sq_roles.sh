# usage: bash tools/sq_roles.sh   (GPU box) -- instruction counts of the lean cfg2 kernel by role: ablation builds with roles removed
# (tools/ablate.py build mdpp_discrete_lean.hip "c:..." "cno12:..." "hno12:..." first), SQ counters per env step of a 64-env wave group
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/sqroles; mkdir -p $o
python3 - <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from mdp_playground_amd import build as B
import ablate
src = "mdpp_discrete_lean.hip"
objs = [os.path.join(ablate.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != src]
for name in sys.argv[1:] or ["c", "cno12", "hno12"]:
    so = os.path.join(ablate.OUT, f"libmdpp__{name}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so] + objs + [ablate.obj_of(src, name)])
PY
for v in shipped c cno12 hno12; do
  lib=build/ablate/libmdpp__$v.so; [ $v = shipped ] && lib=-
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    t=$(echo $c | tr ' ' '_'); rm -rf $o/$v_$t
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $o/${v}_$t -- python3 tools/run_variant.py $lib 2 cfg2 512 > $o/${v}_$t.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob
for v in ("shipped", "c", "cno12", "hno12"):
    tot = {}
    for f in glob.glob(f"gpurun_out/sqroles/{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_lean" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # 2 launches x 512 steps x 1024 wave groups (64 envs: one wave of each role)
    print(v, {k: round(x / (2 * 512 * 1024), 1) for k, x in sorted(tot.items())})
PY
rm -rf $o/*_SQ_*
