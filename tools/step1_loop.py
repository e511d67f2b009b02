"""n single steps (mdpp_step) of a bench workload -- the program behind `rocprofv3 --pmc ... -- python3 tools/step1_loop.py`
(tools/pmc_step1.sh): what a one-step launch issues, per kernel.    python3 tools/step1_loop.py <workload> [n] [rng] [opts,...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = sys.argv[3] if len(sys.argv) > 3 else "numpy"
opts = [o for o in (sys.argv[4].split(",") if len(sys.argv) > 4 else []) if o]
wl = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
env = bench.make_env(wl, wl["envs"], dev, rng)
if opts:
    env.set_kernel_options(*opts)
env.reset()
acts = bench.make_actions(wl, 8, wl["envs"], dev, 1)
for k in range(n):
    env.step(acts[k % 8])
torch.cuda.synchronize()
print("kernel", env.rollout_kernel_name(1), "steps", n, flush=True)
