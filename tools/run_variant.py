#!/usr/bin/env python3
"""Run the cfg2 fused rollout with one ablation build (for rocprofv3 --pmc).  GPU box only.
usage: python3 tools/run_variant.py <so-path> [launches]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mdp_playground_amd import _capi  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _capi.LIB_PATH = os.path.abspath(sys.argv[1])
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
           action_space_size=8, delay=4, sequence_length=3, seed=0)
N, F = 65536, 128
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
acts = torch.randint(0, 8, (F, N), device=env.device, dtype=torch.int32)
out = env.alloc_rollout(F)
for _ in range(n):
    env.rollout(acts, out)
torch.cuda.synchronize()
