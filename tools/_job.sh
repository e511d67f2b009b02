cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r05a; mkdir -p $o
timeout 2400 python3 -m pytest tests/test_gpu_dist.py tests/test_bench_host.py -m gpu -x -q > $o/tests_dist.log 2>&1; echo "dist tests rc=$?"
tail -8 $o/tests_dist.log
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $o/bench.json 2> $o/bench.err
tail -4 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r05a/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        r = d["roofline"]
        print("value", d["value"], "none", d["value_none"], "frac", r["frac"], "replayed", r.get("frac_replayed"), "frac_value", r.get("frac_value"))
        print("peaks", r.get("peak_measured"))
        print({k: r.get(k) for k in ("frac_of_measured_copy", "frac_of_measured_write", "frac_of_measured_read")})
        print("single", json.dumps(d["single_step"])[:1500])
        print("cpu", json.dumps(d["cpu_baseline"])[:800])
        for k, v in (d["workloads"] or {}).items():
            print(k, {kk: v.get(kk) for kk in ("launch_us", "frac", "error")}, json.dumps(v.get("single_step"))[:600])
PY
