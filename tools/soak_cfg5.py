#!/usr/bin/env python3
"""Soak of the numpy-stream noise path of the continuous rollout (GPU box): k_continuous_rollout_fast with its helper wave (batched
ziggurat attempts, parked lanes) against the general kernel on every env, launches of mixed lengths, outputs and the streams'
end states.   python3 tools/soak_cfg5.py [launches]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv, _capi as capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bad = 0
for over in ({}, {"reward_noise": None}, {"transition_noise": None, "reward_noise": 0.3}, {"delay": 3, "max_episode_steps": 9}):
    over = dict(over)
    mes = over.pop("max_episode_steps", None)
    cfg = {k: v for k, v in dict(bench.WORKLOADS["cfg5"]["config"], **over).items() if v is not None}
    N = 32768
    a = RLToyVectorEnv(num_envs=N, autoreset="same_step", max_episode_steps=mes, **cfg)
    b = RLToyVectorEnv(num_envs=N, autoreset="same_step", max_episode_steps=mes, **cfg)
    b.set_kernel_options("NO_CFAST")
    assert a.rollout_kernel_name(64) != b.rollout_kernel_name(64)
    g = torch.Generator(device=a.device); g.manual_seed(4)
    rs = np.random.default_rng(1)
    for j in range(n):
        K = int(rs.choice([64, 33, 100, 37, 48, 32]))
        acts = torch.rand((K, N, 12), generator=g, device=a.device) * 2 - 1
        ra, rb = a.rollout(acts), b.rollout(acts)
        for x, y in zip(ra, rb):
            if not torch.equal(x, y):
                bad += 1
                print("MISMATCH", over, "launch", j, "K", K, flush=True)
    for st in (capi.STREAM_ENV, capi.STREAM_SPACE):
        if not np.array_equal(a.get_rng_streams(st), b.get_rng_streams(st)):
            bad += 1
            print("STREAM MISMATCH", over, st, flush=True)
    print(over, mes, a.rollout_kernel_name(64), "launches", n, "status", int((a.status() != 0).sum()), "bad", bad, flush=True)
    a.close(); b.close()
print("SOAK_OK" if bad == 0 else "SOAK_FAILED")
sys.exit(0 if bad == 0 else 1)
