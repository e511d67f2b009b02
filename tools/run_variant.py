#!/usr/bin/env python3
"""Run fused rollouts of one bench workload with one (ablation) build, for rocprofv3 --pmc.
GPU box only.  usage: python3 tools/run_variant.py <so-path or -> [launches] [workload] [fuse]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mdp_playground_amd import _capi  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _capi.LIB_PATH = os.path.abspath(sys.argv[1])
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wname = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
F = int(sys.argv[4]) if len(sys.argv) > 4 else 128
wl = bench.WORKLOADS[wname]
N = wl["envs"]
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **wl["config"])
acts = bench.make_actions(wl, F, N, env.device, 12345)
out = env.alloc_rollout(F)
for _ in range(n):
    env.rollout(acts, out)
torch.cuda.synchronize()
print(f"workload={wname} envs={N} fuse={F} launches={n}")
