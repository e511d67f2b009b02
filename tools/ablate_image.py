#!/usr/bin/env python3
"""Timing-only ablation builds of k_image_obs (GPU box).  Each variant removes one piece of the
kernel (results are WRONG by construction; only the time matters).  python tools/ablate_image.py
"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
VARIANTS = ["", "NOTPL", "ZERO", "NOSTORE", "ZERO,NOSTORE"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"]


def main():
    from mdp_playground_amd import build as B
    outdir = os.path.join(ROOT, "gpurun_out", "ablate_img")
    os.makedirs(outdir, exist_ok=True)
    objs = [os.path.join(CSRC, os.path.splitext(f)[0] + ".o") for f in B.SOURCES if "image" not in f]
    for v in (sys.argv[1:] or VARIANTS):
        tag = v.replace(",", "_") or "FULL"
        obj = os.path.join(outdir, f"img_{tag}.o")
        so = os.path.join(outdir, f"libmdpp_{tag}.so")
        defs = [("-DMDPP_IMG_SWZ" if d == "SWZ" else f"-DMDPP_IMG_ABL_{d}") for d in v.split(",") if d]
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-c", os.path.join(CSRC, "mdpp_image.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", so] + objs + [obj])
        code = f"""
import sys, time, torch
sys.path.insert(0, {ROOT!r})
from mdp_playground_amd import _capi
_capi.LIB_PATH = {so!r}
from mdp_playground_amd import RLToyVectorEnv
cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
           action_space_size=8, delay=0, seed=0, image_representations=True, image_width=84,
           image_height=84, image_transforms="shift,rotate", image_sh_quant=1, image_ro_quant=1)
N, F = 8192, 16
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
acts = torch.randint(0, 8, (F, N), device=env.device, dtype=torch.int32)
out = env.alloc_rollout(F)
for _ in range(3): env.rollout(acts, out)
torch.cuda.synchronize()
env.timer_begin()
for _ in range(20): env.rollout(acts, out)
ms = env.timer_end()
print("%-32s %8.1f us/step (step kernel + image kernel)" % ({tag!r}, ms*1e3/20/F))
"""
        subprocess.check_call([sys.executable, "-c", code])


if __name__ == "__main__":
    main()
