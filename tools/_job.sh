cd $GRAFT_REPO_ROOT
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/b.json 2> gpurun_out/b.err
tail -4 gpurun_out/b.err
python3 - <<'PY'
import json
for l in open("gpurun_out/b.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("value", d["value"], "frac", d["roofline"]["frac"])
        for k, v in d["workloads"].items():
            print(k, {kk: v.get(kk) for kk in ("launch_us", "frac", "valu_frac", "bound", "traffic", "kernel", "error")})
PY
