#!/usr/bin/env python3
"""Timing-only ablation builds of k_continuous_rollout_fast (GPU box): python tools/ablate_cont.py"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
VARIANTS = ["", "-DMDPP_ABL_NOLOAD", "-DMDPP_ABL_NOSTORE", "-DMDPP_ABL_NOLOAD -DMDPP_ABL_NOSTORE"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]
sys.path.insert(0, ROOT)
from mdp_playground_amd import build as B  # noqa: E402
OBJS = [os.path.splitext(f)[0] + ".o" for f in B.SOURCES if f != "mdpp_continuous_fast.hip"]


def main():
    outdir = os.path.join(ROOT, "gpurun_out", "ablate_c")
    os.makedirs(outdir, exist_ok=True)
    for n, v in enumerate(VARIANTS):
        obj = os.path.join(outdir, f"cf_{n}.o")
        so = os.path.join(outdir, f"libmdpp_c{n}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + v.split() + ["-c", os.path.join(CSRC, "mdpp_continuous_fast.hip"), "-o", obj],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", so] + [os.path.join(CSRC, o) for o in OBJS] + [obj])
        code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from mdp_playground_amd import _capi
_capi.LIB_PATH = {so!r}
from mdp_playground_amd import RLToyVectorEnv
cfg = dict(state_space_type="continuous", state_space_dim=12, relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
           target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10, action_space_max=1,
           transition_dynamics_order=1, inertia=1, time_unit=1, make_denser=True, reward_function="move_to_a_point", seed=0)
N, F = 65536, 512
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
acts = (torch.rand((F, N, 12), device=env.device) * 2 - 1)
out = env.alloc_rollout(F)
for _ in range(3): env.rollout(acts, out)
torch.cuda.synchronize()
env.timer_begin()
for _ in range(20): env.rollout(acts, out)
ms = env.timer_end()
print("%-28s %8.1f us/launch  %6.0f ns/step  %.0f GB/s alg" % ({v!r} or "FULL", ms*1e3/20, ms*1e6/20/F, 102*N*F/(ms/20*1e-3)/1e9))
"""
        subprocess.check_call([sys.executable, "-c", code])


if __name__ == "__main__":
    main()
