"""Where the host's time per mdpp_step call goes (eager single steps are host-bound below about 3.5 us per launch).
    python tools/host_cost.py [cfg2]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402
from mdp_playground_amd import RLToyVectorEnv             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[name]
N = wl["envs"]
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **wl["config"])
env.reset()
dev = env.device
g = np.random.default_rng(0)
if wl["kind"] == "discrete":
    a1 = torch.as_tensor(g.integers(0, 8, size=(N,)).astype(np.int32), device=dev)
else:
    a1 = torch.as_tensor(g.uniform(-1, 1, size=(N, env.mdps[0].D)).astype(np.float32), device=dev)
n = 2000


def t(f, label):
    for _ in range(100):
        f()
    torch.cuda.synchronize(dev)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        dt = (time.perf_counter() - t0) * 1e6 / n
        torch.cuda.synchronize(dev)
        best = min(best, dt)
    print(json.dumps({"host_cost": label, "us_per_call": round(best, 3)}), flush=True)


t(lambda: None, "empty lambda")
t(lambda: torch.cuda.current_stream(dev).cuda_stream, "torch.cuda.current_stream(dev).cuda_stream")
if hasattr(torch._C, "_cuda_getCurrentRawStream"):
    t(lambda: torch._C._cuda_getCurrentRawStream(dev.index), "torch._C._cuda_getCurrentRawStream")
t(lambda: a1.data_ptr(), "tensor.data_ptr()")
t(lambda: (torch.is_tensor(a1) and a1.dtype == env._act_dtype and a1.device == dev and a1.shape == env._act_shape and a1.is_contiguous()),
  "the action tensor checks")
lib, h = env._lib, env._h
args = (h, a1.data_ptr(), env._p_obs, env._p_reward, env._p_term, env._p_trunc, env._p_final, torch.cuda.current_stream(dev).cuda_stream)
step = env._mdpp_step
t(lambda: step(*args), "ctypes mdpp_step, prebuilt arguments")
t(lambda: env.step(a1), "env.step(a1)")
tick = C.c_uint64()
t(lambda: lib.mdpp_tick(h, 0, C.byref(tick)), "ctypes call of a trivial function (mdpp_tick)")
