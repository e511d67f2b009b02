import sys, torch
sys.path.insert(0, ".")
import bench
from mdp_playground_amd import mdp as M, RLToyVectorEnv
dev = torch.device("cuda", 0)
base = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8)
for tag, cfg in (("L3_delay4", dict(base, delay=4, sequence_length=3)), ("L1_delay4_sf", dict(base, delay=4, sequence_length=1)),
                 ("L1_rn0", dict(base, delay=0, sequence_length=1, reward_noise=0.0))):
    N = 65536
    env = RLToyVectorEnv(seeds=list(range(N)), device=dev, autoreset="same_step", **cfg)
    wl = dict(kind="discrete", config=cfg)
    acts = bench.action_rotation(wl, 512, N, dev, 1)
    out = env.alloc_rollout(512)
    for i in range(3): env.rollout(acts[i % len(acts)], out)
    torch.cuda.synchronize()
    env.timer_begin()
    for i in range(10): env.rollout(acts[i % len(acts)], out)
    ms = env.timer_end()
    print(tag, env.rollout_kernel_name(512), "launch_us %.1f frac %.3f" % (ms * 100, 18 * N * 512 / (ms * 1e-4) / 8e12), flush=True)
    env.close()
