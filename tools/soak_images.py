#!/usr/bin/env python3
"""Soak of the image pipeline at the bench size (GPU box): the fast renderer (waves claim their images from counters, batches
of 64 steps on two streams) against the general renderer on every pixel, many rollouts of mixed lengths, outputs poisoned
before every launch (an image nobody claimed stays poisoned).   python3 tools/soak_images.py [rollouts]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bench.WORKLOADS["cfg4"]
N = wl["envs"]
a = RLToyVectorEnv(num_envs=N, autoreset="same_step", **wl["config"])
b = RLToyVectorEnv(num_envs=N, autoreset="same_step", **wl["config"])
b.set_kernel_options("NO_IMGFAST")
rs = np.random.default_rng(0)
bad = 0
for j in range(n):
    K = int(rs.choice([300, 130, 64, 65, 17, 200]))
    acts = bench.make_actions(wl, K, N, a.device, 500 + j)
    oa = a.alloc_rollout(K); ob = b.alloc_rollout(K)
    oa[0].fill_(0x5A); ob[0].fill_(0xA5)
    ra = a.rollout(acts, oa); rb = b.rollout(acts, ob)
    torch.cuda.synchronize()
    for k in range(K):
        if not torch.equal(ra[0][k], rb[0][k]):
            bad += 1
            print("MISMATCH rollout", j, "K", K, "step", k, flush=True)
            break
    ok = torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])
    bad += 0 if ok else 1
    print("rollout", j, "K", K, "ok" if bad == 0 else "BAD", flush=True)
    del oa, ob, ra, rb
print("SOAK_OK" if bad == 0 else "SOAK_FAILED")
sys.exit(0 if bad == 0 else 1)
