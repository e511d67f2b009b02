timeout 1500 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/t6.log 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-workloads --no-single-step 2>/dev/null | grep '^{"metric"' > gpurun_out/bench_peer.json
tail -n 12 gpurun_out/t6.log
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_peer.json").read())
print({k:(v.get("env_steps_per_s"), v.get("error"), v.get("timeouts")) for k,v in d["collective_legs"].items()})
print({k:v for k,v in d["multi_rank_diagnostics"].items() if k not in ("per_rank","note")})
PY
