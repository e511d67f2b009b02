cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step1" 2>&1 | tail -4
python3 - <<'PY'
import sys, json; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, bench
import bench_step1 as b1
wl = bench.WORKLOADS["cfg2_noise"]
b1.timing("cfg2_noise", wl, wl["envs"], "numpy", ("NO_STEP1",), "general kernel")
b1.timing("cfg2_noise", wl, wl["envs"], "numpy", (), "default")
b1.timing("cfg2_noise", wl, wl["envs"], "philox", ("NO_STEP1",), "general kernel")
b1.timing("cfg2_noise", wl, wl["envs"], "philox", (), "default")
PY
