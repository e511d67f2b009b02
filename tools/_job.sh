cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, bench
N = 65536
for label, over in (("c_d2_n0", {}), ("order 2", {"transition_dynamics_order": 2}), ("order 3", {"transition_dynamics_order": 3}), ("sigma 0.5 / 1", {"transition_noise": 0.5, "reward_noise": 1})):
    wl = dict(bench.WORKLOADS["c_d2_n0"]); wl["config"] = dict(wl["config"], **over)
    for opts in ((), ("NO_TRIO",)):
        env = bench.make_env(wl, N, torch.device("cuda", 0), "numpy")
        if opts: env.set_kernel_options(*opts)
        env.reset()
        acts = [bench.make_actions(wl, 512, N, env.device, 1 + j) for j in range(3)]
        out = env.alloc_rollout(512)
        for k in range(3): env.rollout(acts[k % 3], out)
        torch.cuda.synchronize(); best = 1e9
        for rep in range(3):
            env.timer_begin()
            for k in range(6): env.rollout(acts[k % 3], out)
            ms = env.timer_end(); torch.cuda.synchronize(); best = min(best, ms * 1e3 / 6)
        print("%-16s %-10s rollout %7.1f us  %s" % (label, opts, best, env.rollout_kernel_name(512)), flush=True)
        env.close()
PY
timeout 1200 python3 -m pytest tests/test_gpu_sweep.py -m gpu -q -k "c_x" 2>&1 | tail -4
