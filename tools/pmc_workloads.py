#!/usr/bin/env python3
"""Child of bench.py's live PMC passes (GPU box only): fused rollouts of several bench workloads in ONE
process, for `rocprofv3 --pmc <counter> -- python3 tools/pmc_workloads.py <launches> <name>:<rng>:<envs>:<fuse> ...`.
After the launches of every workload ONE marker kernel is dispatched (mdpp_philox_normals, one lane), so that the
counter rows, read in dispatch order, split into one segment per workload."""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv, _capi  # noqa: E402
import bench  # noqa: E402

import time  # noqa: E402
launches = int(sys.argv[1])
t_start = time.perf_counter()
lib = _capi.load()
dev = torch.device("cuda", 0)
mark = torch.zeros(4, dtype=torch.float32, device=dev)
for spec in sys.argv[2:]:
    wname, rng, envs, fuse = spec.split(":")
    wl = bench.WORKLOADS[wname]
    N, F = int(envs), int(fuse)
    t0 = time.perf_counter()
    env = bench.make_env(wl, N, dev, rng)
    t1 = time.perf_counter()
    acts = bench.make_actions(wl, F, N, dev, 12345)
    out = env.alloc_rollout(F)
    for _ in range(launches):
        env.rollout(acts, out)
    torch.cuda.synchronize()
    print(f"workload={wname} rng={rng} envs={N} fuse={F} launches={launches} kernel={env.rollout_kernel_name(F)} "
          f"make_env_s={t1 - t0:.2f} total_s={time.perf_counter() - t0:.2f} since_start_s={time.perf_counter() - t_start:.2f}", flush=True)
    env.close()
    del env, acts, out
    lib.mdpp_philox_normals(1, 0, 0, 0, 1, 1, C.c_void_p(mark.data_ptr()),
                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
