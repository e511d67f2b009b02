#!/usr/bin/env python3
"""Time the cfg2 fused rollout of ONE source tree (GPU box): replayed and rotating action tensors, R repeats of L launches.
usage: python3 tools/bisect_lean.py <tree root> [repeats] [launches] [fuse] [rng]
Used to set two trees (round 2's HEAD extracted under build/r2tree, and this one) side by side on one lease:
the same script, the same tensors, alternating processes (tools/bisect_lean.sh)."""
import json
import os
import statistics
import sys

tree = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else ".")
sys.path.insert(0, tree)
import torch  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402

R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20
F = int(sys.argv[4]) if len(sys.argv) > 4 else 512
rng = sys.argv[5] if len(sys.argv) > 5 else "numpy"
N = 65536
cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
           delay=4, sequence_length=3, seed=0)
kw = dict(rng=rng, philox_seed=12345) if rng == "philox" else {}
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg, **kw)
g = torch.Generator(device=env.device)
g.manual_seed(12345)
acts = [torch.randint(0, 8, (F, N), dtype=torch.int32, device=env.device, generator=g) for _ in range(4)]
out = env.alloc_rollout(F)
for a in acts:
    env.rollout(a, out)
torch.cuda.synchronize()


def leg(rot):
    us = []
    for _ in range(R):
        env.timer_begin()
        for j in range(L):
            env.rollout(acts[j % 4] if rot else acts[0], out)
        us.append(env.timer_end() * 1e3 / L)
    return us


res = {"tree": tree, "kernel": env.rollout_kernel_name(F), "rng": rng}
for name, rot in (("replayed", False), ("rotating", True), ("replayed2", False), ("rotating2", True)):
    us = leg(rot)
    res[name] = {"min": round(min(us), 2), "median": round(statistics.median(us), 2), "max": round(max(us), 2)}
print(json.dumps(res))
