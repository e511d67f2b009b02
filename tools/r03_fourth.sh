cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03d; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_post.py -m gpu -x -q -k "default_target or continuous or post or episode" > $o/tests_a.log 2>&1; echo "rc=$?" >> $o/tests_a.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $o/bench.json 2> $o/bench.err
tail -5 $o/tests_a.log; tail -c 400 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r03d/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("value", d["value"], "none", d["value_none"], "last", d["value_last_row"])
        r = d["roofline"]; print("frac", r["frac"], "replayed", r["frac_replayed"], "launch_us", r["launch_us"], r["launch_us_replayed"], "traffic", r["traffic"])
        print({k: {kk: vv for kk, vv in v.items() if kk in ("elapsed_s", "host_enqueue_s")} for k, v in d["collective_legs"].items()})
        for k, v in (d["workloads"] or {}).items():
            print(k, {kk: v.get(kk) for kk in ("env_steps_per_s", "launch_us", "frac", "traffic", "error")})
PY
