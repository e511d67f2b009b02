#!/usr/bin/env python3
"""Soak of the noise kernels (GPU box): k_discrete_rollout_lean<..., PN, RN> against the quiet kernel and the general kernel
on EVERY env of the bench shape, many launches of mixed lengths (a hand-off race between the role waves shows up as a rare
single-lane difference); on numpy streams also the end states of both streams of every env.
python3 tools/soak_noise.py [launches] [philox|numpy]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng_mode = sys.argv[2] if len(sys.argv) > 2 else "philox"
N = 65536
bad = 0
for over in ({}, {"reward_noise": None}, {"transition_noise": None}, {"max_episode_steps": 11}, {"reward_every_n_steps": 3, "delay": 0}):
    over = dict(over)
    mes = over.pop("max_episode_steps", None)
    cfg = {k: v for k, v in dict(bench.WORKLOADS["cfg2_noise"]["config"], **over).items() if v is not None}
    kw = dict(rng="philox", philox_seed=7) if rng_mode == "philox" else {}
    envs = [RLToyVectorEnv(num_envs=N, autoreset="same_step", max_episode_steps=mes, **kw, **cfg) for _ in range(3)]
    envs[1].set_kernel_options("NO_LEAN")
    envs[2].set_kernel_options("NO_PHILOX_FAST" if rng_mode == "philox" else "NO_LEAN", *([] if rng_mode == "philox" else ["NO_QUIET", "NO_QUIET_NOISE"]))
    names = [e.rollout_kernel_name(512) for e in envs]
    assert len(set(names)) == 3, names
    g = torch.Generator(device=envs[0].device)
    g.manual_seed(1)
    rs = np.random.default_rng(0)
    for j in range(launches):
        K = int(rs.choice([512, 512, 200, 37, 64, 33]))
        acts = torch.randint(0, 8, (K, N), generator=g, device=envs[0].device, dtype=torch.int32)
        outs = [e.rollout(acts) for e in envs]
        for k in (1, 2):
            for x, y in zip(outs[0], outs[k]):
                if not torch.equal(x, y):
                    bad += 1
                    d = (x != y).nonzero()
                    print("MISMATCH", over, "launch", j, "K", K, "vs", names[k], "first", d[:4].tolist(), flush=True)
    if rng_mode == "numpy":
        for stream in (0, 1):
            ref = envs[0].get_rng_streams(stream)
            for k in (1, 2):
                if not np.array_equal(ref, envs[k].get_rng_streams(stream)):
                    bad += 1
                    print("STREAM MISMATCH", over, "stream", stream, "vs", names[k], flush=True)
    st = [int((e.status() != 0).sum()) for e in envs]
    print(over, mes, names[0], "launches", launches, "status bits", st, "mismatches so far", bad, flush=True)
    for e in envs:
        e.close()
print("SOAK_OK" if bad == 0 else "SOAK_FAILED")
sys.exit(0 if bad == 0 else 1)
