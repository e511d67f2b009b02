#!/usr/bin/env python3
"""Run fused rollouts of one bench workload with one (ablation) build, for rocprofv3 --pmc.
GPU box only.  usage: python3 tools/run_variant.py <so-path or -> [launches] [workload] [fuse] [rng=philox]
[disable=NO_CFAST,NO_PARK] [envs=N]   (disable: mdpp_set_options switches, include/mdpp.h MDPP_OPT_*)"""
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

kw = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
pos = [a for a in sys.argv[2:] if "=" not in a]
n = int(pos[0]) if len(pos) > 0 else 10
wname = pos[1] if len(pos) > 1 else "cfg2"
F = int(pos[2]) if len(pos) > 2 else 128
wl = bench.WORKLOADS[wname]
N = int(kw.get("envs", wl["envs"]))
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", rng=kw.get("rng", "numpy"), **wl["config"])
if kw.get("disable"):
    env.set_kernel_options(*kw["disable"].split(","))
print("kernel:", env.rollout_kernel_name(F))
acts = bench.make_actions(wl, F, N, env.device, 12345)
out = env.alloc_rollout(F)
for _ in range(n):
    env.rollout(acts, out)
torch.cuda.synchronize()
print(f"workload={wname} envs={N} fuse={F} launches={n}")
