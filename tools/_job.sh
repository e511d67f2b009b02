cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py -m gpu -q -x -k "d_x or quiet or specialised or s24 or s50 or noise" 2>&1 | tail -5
python3 - <<'PY'
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, bench
N = 65536
base = bench.WORKLOADS["d_s8_rn0"]
for label, over in (("S=20 rn0", {"state_space_size": 20, "action_space_size": 20}), ("S=50 rn0", {"state_space_size": 50, "action_space_size": 50}),
                    ("S=50 rn0 delay 4", {"state_space_size": 50, "action_space_size": 50, "delay": 4}), ("S=20 pn0.1 rn0", {"state_space_size": 20, "action_space_size": 20, "transition_noise": 0.1}),
                    ("S=24 rdist rn0", {"state_space_size": 24, "action_space_size": 24, "reward_dist": [0.01, 1]})):
    cfg = {k: v for k, v in dict(base["config"], **over).items() if v is not None}
    wl = dict(base, config=cfg)
    env = bench.make_env(wl, N, torch.device("cuda", 0), "numpy"); env.reset()
    acts = [bench.make_actions(wl, 512, N, env.device, 1 + j) for j in range(3)]
    out = env.alloc_rollout(512)
    for k in range(3): env.rollout(acts[k % 3], out)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(3):
        env.timer_begin()
        for k in range(6): env.rollout(acts[k % 3], out)
        ms = env.timer_end(); torch.cuda.synchronize(); best = min(best, ms * 1e3 / 6)
    print("%-20s rollout %7.1f us  %s" % (label, best, env.rollout_kernel_name(512)), flush=True)
    env.close()
PY
