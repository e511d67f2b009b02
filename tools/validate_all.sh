# usage: bash tools/validate_all.sh   (GPU box) -- what the driver runs at round end, in one call: the -m gpu suite, smoke(), the bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/validate; mkdir -p $o
timeout 1800 python3 -m pytest tests -m gpu -x -q > $o/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $o/tests_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $o/smoke.log 2>&1; echo "smoke rc=$?" >> $o/smoke.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $o/bench.json 2> $o/bench.err
tail -4 $o/tests_gpu.log; tail -3 $o/smoke.log; tail -4 $o/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/validate/bench.json"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("value", d["value"], "none", d["value_none"], "frac", d["roofline"]["frac"], "replayed", d["roofline"]["frac_replayed"])
        for k, v in (d["workloads"] or {}).items():
            print(k, {kk: v.get(kk) for kk in ("launch_us", "frac", "error")})
PY
