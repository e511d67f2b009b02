#!/usr/bin/env python3
"""Timing-only ablation builds of k_discrete_rollout_fast (GPU box).  Each variant removes one
piece of the step (results are WRONG by construction; only the launch time matters) to show
where the per-step time goes.  Usage on the GPU box:  python tools/ablate_fast.py
"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
VARIANTS = ["", "NOSTORE", "NORESET", "NOSTORE,NORESET"]
OLD_VARIANTS = ["", "HALFWAVE", "HALFWAVE,NOSTORE", "NOSTORE", "NORESET", "NOREFILL", "NOLDSR", "NOLDSP", "NOLDSR,NOLDSP",
            "NOSTORE,NORESET,NOREFILL", "NOSTORE,NORESET,NOREFILL,NOLDSR,NOLDSP"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]


def main():
    import torch
    outdir = os.path.join(ROOT, "gpurun_out", "ablate")
    os.makedirs(outdir, exist_ok=True)
    from mdp_playground_amd import build as B
    objs = [os.path.join(CSRC, os.path.splitext(f)[0] + ".o") for f in B.SOURCES if f != "mdpp_discrete_pipe.hip"]
    for v in VARIANTS:
        tag = v.replace(",", "_") or "FULL"
        obj = os.path.join(outdir, f"fast_{tag}.o")
        so = os.path.join(outdir, f"libmdpp_{tag}.so")
        defs = [f"-DMDPP_ABL_{d}" for d in v.split(",") if d]
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-c", os.path.join(CSRC, "mdpp_discrete_pipe.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", so] + objs + [obj])
        code = f"""
import sys, time, torch
sys.path.insert(0, {ROOT!r})
from mdp_playground_amd import _capi
_capi.LIB_PATH = {so!r}
from mdp_playground_amd import RLToyVectorEnv
cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
           action_space_size=8, delay=4, sequence_length=3, seed=0)
N, F = 65536, 512
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
acts = torch.randint(0, 8, (F, N), device=env.device, dtype=torch.int32)
out = env.alloc_rollout(F)
for _ in range(5): env.rollout(acts, out)
torch.cuda.synchronize()
env.timer_begin()
for _ in range(40): env.rollout(acts, out)
ms = env.timer_end()
print("%-45s %8.1f us/launch  %6.0f ns/step" % ({tag!r}, ms*1e3/40, ms*1e6/40/F))
"""
        subprocess.check_call([sys.executable, "-c", code])


if __name__ == "__main__":
    main()
