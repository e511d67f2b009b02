cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $o/bench.json 2> $o/bench.err
tail -4 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r05b/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        r = d["roofline"]
        print("value", d["value"], "frac", r["frac"], "valu", r.get("valu"), "traffic", r.get("traffic"), r.get("traffic_source"))
        for k, v in (d["workloads"] or {}).items():
            print(k, {kk: v.get(kk) for kk in ("launch_us", "frac", "valu_frac", "bound", "traffic", "alg_bytes_per_launch", "kernel", "error")})
PY
