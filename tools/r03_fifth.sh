cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03e; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "default_target or line or rccl or continuous" > $o/tests_a.log 2>&1; echo "rc=$?" >> $o/tests_a.log
python3 tools/ablate.py run mdpp_continuous.hip line numpy shipped > $o/line.txt 2>&1
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-workloads --no-single-step ) > $o/bench.json 2> $o/bench.err
tail -5 $o/tests_a.log; cat $o/line.txt; tail -c 300 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r03e/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("value", d["value"], "none", d["value_none"], "last", d["value_last_row"])
        print({k: {kk: vv for kk, vv in v.items() if kk in ("elapsed_s", "host_enqueue_s")} for k, v in d["collective_legs"].items()})
        print(d["config"]["collective"])
PY
