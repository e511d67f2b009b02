#!/usr/bin/env python3
"""HIP-event time per fused launch of bench.py workloads, with optional dispatch switches: a quick A/B on one lease.

    python3 tools/time_legs.py d_s50_delay4 d_s24_rdist:NO_QUIET_SF cfg2_per_env c_d2_n0:NO_SIGMA0 [--rng philox] [--reps 3]
"""
import os
import statistics
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

rng = sys.argv[sys.argv.index("--rng") + 1] if "--rng" in sys.argv else "numpy"
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
args, skip = [], False
for a in sys.argv[1:]:
    if skip:
        skip = False
    elif a in ("--rng", "--reps"):
        skip = True
    else:
        args.append(a)
dev = torch.device("cuda", 0)
for spec in args:
    name, _, opts = spec.partition(":")
    wl = bench.WORKLOADS[name]
    N = wl["envs"]
    F = max(1, min(512, wl.get("fuse_max", 512)))
    env = bench.make_env(wl, N, dev, rng)
    if opts:
        env.set_kernel_options(*opts.split(","))
    acts = bench.action_rotation(wl, F, N, dev, 12345)
    out = env.alloc_rollout(F)
    for i in range(3):
        env.rollout(acts[i % len(acts)], out)
    torch.cuda.synchronize()
    us = []
    n = 3
    for _ in range(reps):
        env.timer_begin()
        for i in range(10):
            env.rollout(acts[(n + i) % len(acts)], out)
        us.append(env.timer_end() * 100.0)
        n += 10
    med = statistics.median(us)
    frac = wl["alg_bytes_fused"] * N * F / (med * 1e-6) / 8e12
    print(f"{spec:28s} {env.rollout_kernel_name(F):110s} launch_us {med:8.1f} (min {min(us):.1f}) frac {frac:.3f} status {int((env.status() != 0).sum())}", flush=True)
    env.close()
    del env, acts, out
    torch.cuda.empty_cache()
