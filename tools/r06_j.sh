cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/ablate.py run mdpp_image.hip img100_all numpy wide0 all1 wide0 all1 2>&1 | grep -v amdgpu.ids
python3 tools/ablate.py run mdpp_image.hip img100_shift numpy wide0 all1 2>&1 | grep -v amdgpu.ids
python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for name in ("cfg4", "img100_all"):
    wl = bench.WORKLOADS[name]
    env = bench.make_env(wl, wl["envs"], dev, "numpy")
    acts = bench.make_actions(wl, 64, wl["envs"], dev, 1)
    r = bench.single_step_leg(env, wl, acts, wl["envs"], dev, n1=300, reps=5)
    print(name, "step us", r["launch_us_events"], "graph", (r.get("graph") or {}).get("us_per_step_events"))
    env.close()
PY
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "image or cfg4 or img" 2>&1 | tail -3
