# round 3, first GPU call: new boundary / dist / cfg4 tests, then the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03a; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_boundary.py -m gpu -x -q > $o/tests_boundary.log 2>&1; echo "boundary rc=$?" >> $o/tests_boundary.log
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg4_at_bench_size" > $o/tests_cfg4.log 2>&1; echo "cfg4 rc=$?" >> $o/tests_cfg4.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $o/bench.json 2> $o/bench.err
tail -5 $o/tests_boundary.log; tail -5 $o/tests_cfg4.log; tail -c 600 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r03a/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("value", d["value"], "none", d["value_none"], "last", d["value_last_row"])
        r = d["roofline"]; print("frac", r["frac"], "replayed", r["frac_replayed"], "launch_us", r["launch_us"], r["launch_us_replayed"], "traffic", r["traffic"], r["traffic_source"])
        print(d["config"]["collective"])
        for k, v in (d["workloads"] or {}).items():
            print(k, {kk: v.get(kk) for kk in ("env_steps_per_s", "launch_us", "frac", "traffic", "kernel", "error")})
PY
