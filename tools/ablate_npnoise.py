#!/usr/bin/env python3
"""Timing-only ablation builds of k_discrete_rollout_lean<PHILOX=0, PN, RN> (cfg2 + noise on numpy streams):
    python3 tools/ablate_npnoise.py build [n ...]   (here)        python3 tools/ablate_npnoise.py [n ...]   (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ablate_npnoise")
VARIANTS = [
    ("as shipped", ""),
    ("start states without the cdf search", "-DMDPP_ABL_NP_NOSS"),
    ("transition-noise bytes without the threshold counts", "-DMDPP_ABL_NP_NOPNC"),
    ("every word accepted (no wedge / tail pass)", "-DMDPP_ABL_NP_NOSLOW"),
    ("all three", "-DMDPP_ABL_NP_NOSS -DMDPP_ABL_NP_NOPNC -DMDPP_ABL_NP_NOSLOW"),
    ("prio E 3 O 2 H 3", "-DMDPP_LEAN_PRIO_E=3 -DMDPP_LEAN_PRIO_O=2 -DMDPP_LEAN_PRIO_H=3"),
    ("prio E 1 O 2 H 3", "-DMDPP_LEAN_PRIO_E=1 -DMDPP_LEAN_PRIO_O=2 -DMDPP_LEAN_PRIO_H=3"),
    ("prio E 2 O 1 H 3", "-DMDPP_LEAN_PRIO_E=2 -DMDPP_LEAN_PRIO_O=1 -DMDPP_LEAN_PRIO_H=3"),
    ("prio E 0 O 0 H 0", "-DMDPP_LEAN_PRIO_E=0 -DMDPP_LEAN_PRIO_O=0 -DMDPP_LEAN_PRIO_H=0"),
    ("prio E 3 O 1 H 2", "-DMDPP_LEAN_PRIO_E=3 -DMDPP_LEAN_PRIO_O=1 -DMDPP_LEAN_PRIO_H=2"),
    ("prio E 2 O 0 H 3", "-DMDPP_LEAN_PRIO_E=2 -DMDPP_LEAN_PRIO_O=0 -DMDPP_LEAN_PRIO_H=3"),
    ("prio E 3 O 0 H 3", "-DMDPP_LEAN_PRIO_E=3 -DMDPP_LEAN_PRIO_O=0 -DMDPP_LEAN_PRIO_H=3"),
    ("prio E 1 O 0 H 3, O2 2", "-DMDPP_LEAN_PRIO_E=1 -DMDPP_LEAN_PRIO_O=0 -DMDPP_LEAN_PRIO_H=3 -DMDPP_LEAN_PRIO_O2=2"),
    ("prio E 2 O 2 H 3", "-DMDPP_LEAN_PRIO_E=2 -DMDPP_LEAN_PRIO_O=2 -DMDPP_LEAN_PRIO_H=3"),
    ("O2 at priority 2", "-DMDPP_LEAN_PRIO_O2=2"),
    ("O2 at priority 1", "-DMDPP_LEAN_PRIO_O2=1"),
    ("E 3, O1 1, O2 3, H 3", "-DMDPP_LEAN_PRIO_E=3 -DMDPP_LEAN_PRIO_O=1 -DMDPP_LEAN_PRIO_H=3 -DMDPP_LEAN_PRIO_O2=3"),
    ("E 1, O1 0, O2 3, H 3", "-DMDPP_LEAN_PRIO_E=1 -DMDPP_LEAN_PRIO_O=0 -DMDPP_LEAN_PRIO_H=3 -DMDPP_LEAN_PRIO_O2=3"),
    ("E 2, O1 1, O2 2, H 3", "-DMDPP_LEAN_PRIO_E=2 -DMDPP_LEAN_PRIO_O=1 -DMDPP_LEAN_PRIO_H=3 -DMDPP_LEAN_PRIO_O2=2"),
    ("E 2, O1 1, O2 3, H 2", "-DMDPP_LEAN_PRIO_E=2 -DMDPP_LEAN_PRIO_O=1 -DMDPP_LEAN_PRIO_H=2 -DMDPP_LEAN_PRIO_O2=3"),
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DMDPP_LEAN_SHAPES_MIN"]
sys.path.insert(0, ROOT)
from mdp_playground_amd import build as B  # noqa: E402
SRC = "mdpp_discrete_lean_npnoise.hip"
OBJS = [os.path.splitext(f)[0] + ".o" for f in B.SOURCES if f != SRC]


def build(sel=None):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)

    def one(n):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + VARIANTS[n][1].split() +
                              ["-c", os.path.join(CSRC, SRC), "-o", os.path.join(OUT, f"np_{n}.o")], stdout=subprocess.DEVNULL)
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(one, [n for n in range(len(VARIANTS)) if sel is None or n in sel]))


def run(sel=None):
    for n, (name, v) in enumerate(VARIANTS):
        obj = os.path.join(OUT, f"np_{n}.o")
        if not os.path.exists(obj) or (sel is not None and n not in sel):
            continue
        so = os.path.join("/tmp", f"libmdpp_n{n}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", so] + [os.path.join(CSRC, o) for o in OBJS] + [obj])
        code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from mdp_playground_amd import _capi
_capi.LIB_PATH = {so!r}
from mdp_playground_amd import RLToyVectorEnv
import bench
wl = bench.WORKLOADS["cfg2_noise"]
N, F = 65536, 512
res = []
for over in ({{}}, {{"reward_noise": None}}, {{"transition_noise": None}}):
    cfg = {{k: v for k, v in dict(wl["config"], **over).items() if v is not None}}
    env = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
    assert "lean" in env.rollout_kernel_name(F), env.rollout_kernel_name(F)
    acts = bench.action_rotation(wl, F, N, env.device, 12345)
    out = env.alloc_rollout(F)
    for j in range(2): env.rollout(acts[j], out)
    torch.cuda.synchronize()
    us = []
    for r in range(3):
        env.timer_begin()
        for j in range(5): env.rollout(acts[j % len(acts)], out)
        us.append(env.timer_end() * 1e3 / 5)
    res.append(sorted(us)[1])
    env.close()
print("%-52s %-50s  PN+RN %7.1f   PN %7.1f   RN %7.1f us/launch" % ({name!r}, {v!r}, res[0], res[1], res[2]), flush=True)
"""
        subprocess.run([sys.executable, "-c", code], timeout=300)
        os.remove(so)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "build":
        build([int(x) for x in args[1:]] or None)
    else:
        run([int(x) for x in args] or None)
