cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "c_x or continuous or cfg3 or cfg5 or c_" 2>&1 | tail -3
python3 - <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, bench
for rng in ("numpy", "philox"):
    for over in ({}, {"transition_dynamics_order": 3}):
        wl = dict(bench.WORKLOADS["c_d2_n0"]); wl["config"] = dict(wl["config"], **over); N = wl["envs"]
        env = bench.make_env(wl, N, torch.device("cuda", 0), rng); env.reset()
        acts = [bench.make_actions(wl, 512, N, env.device, 1 + j) for j in range(3)]
        out = env.alloc_rollout(512)
        for k in range(3): env.rollout(acts[k % 3], out)
        torch.cuda.synchronize(); best = 1e9
        for rep in range(3):
            env.timer_begin()
            for k in range(6): env.rollout(acts[k % 3], out)
            ms = env.timer_end(); torch.cuda.synchronize(); best = min(best, ms * 1e3 / 6)
        print(rng, over, "%.1f us" % best, env.rollout_kernel_name(512), flush=True)
        env.close()
PY
