#!/usr/bin/env python3
"""Bench-size soak of the reference's sweep shapes (bench workloads d_s8_rn0, c_d2_n0, img100_all, img100_shift): the default
dispatch against a handle with every specialisation switched off, every env of the bench shape, fused launches and single steps,
every output and every stream's end state bit for bit.    python3 tools/soak_sweep_shapes.py [workload ...]   (GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from mdp_playground_amd import _capi as capi                   # noqa: E402


def same(x, y):
    return torch.equal(x.view(torch.int32) if x.dtype.is_floating_point else x, y.view(torch.int32) if y.dtype.is_floating_point else y)


dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["d_s8_rn0", "c_d2_n0", "img100_all", "img100_shift"]):
    wl = bench.WORKLOADS[name]
    image = bool(wl["config"].get("image_representations"))
    N, F = wl["envs"], (32 if image else 512)
    for rng in ("numpy", "philox"):
        a, b = bench.make_env(wl, N, dev, rng), bench.make_env(wl, N, dev, rng)
        b.set_kernel_options(*capi.OPTIONS)
        a.reset(); b.reset()
        for j in range(2):
            acts = bench.make_actions(wl, F, N, dev, 31 + j)
            ra, rb = a.rollout(acts), b.rollout(acts)
            torch.cuda.synchronize()
            assert all(same(x, y) for x, y in zip(ra, rb)), (name, rng, "rollout", j)
            del ra, rb
            for t in range(8):
                sa, sb = a.step(acts[t]), b.step(acts[t])
                assert all(same(x, y) for x, y in zip(sa[:4], sb[:4])), (name, rng, "step", j, t)
        if rng == "numpy":
            for s in (capi.STREAM_ENV, capi.STREAM_SPACE) + ((capi.STREAM_IMAGE,) if image else ()):
                assert np.array_equal(a.get_rng_streams(s), b.get_rng_streams(s)), (name, s)
        assert np.array_equal(a.status(), b.status())
        print("soak", name, rng, "N", N, "F", F, a.rollout_kernel_name(F), "==", b.rollout_kernel_name(F), "OK", flush=True)
        a.close(); b.close()
print("SOAK_OK")
