#!/usr/bin/env python3
"""Time fused rollouts of an arbitrary config (GPU box): env-steps/s and us per env-wave step.
usage: python3 tools/time_config.py '<json config overrides>' [envs] [fuse] [launches] [base workload]
The overrides are applied to the config of bench.py's base workload (default cfg2)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402
import bench  # noqa: E402

over = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L = int(sys.argv[4]) if len(sys.argv) > 4 else 10
base = bench.WORKLOADS[sys.argv[5] if len(sys.argv) > 5 else "cfg2"]
cfg = dict(base["config"], **over)
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", rng=os.environ.get("MDPP_RNG", "numpy"), **cfg)
wl = dict(base, config=cfg)
acts = bench.make_actions(wl, F, N, env.device, 1)
out = env.alloc_rollout(F)
for _ in range(3):
    env.rollout(acts, out)
torch.cuda.synchronize()
env.timer_begin()
for _ in range(L):
    env.rollout(acts, out)
ms = env.timer_end()
per_launch = ms / L
print(f"{json.dumps(over):70s} {env.rollout_kernel_name(F):28s} {N * F / (per_launch * 1e-3):.3e} env-steps/s "
      f"{per_launch * 1e3 / F:.3f} us/step")
