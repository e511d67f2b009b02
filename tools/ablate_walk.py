#!/usr/bin/env python3
"""Timing-only ablation builds of the cfg5 numpy-stream kernel (generator / walker / consumer waves):
    python3 tools/ablate_walk.py build     (here: compiles the variants of mdpp_continuous_fast.hip into build/ablate_walk/)
    python3 tools/ablate_walk.py           (GPU box: links each variant against the shipped objects and times cfg5)
Results of the ablated variants are garbage by construction; only the launch time is read."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ablate_walk")
VARIANTS = [
    ("as shipped", ""),
    ("consumer does not wait for normals", "-DMDPP_ABL_WK_NOWAIT"),
    ("generator + walker only (no consumer)", "-DMDPP_ABL_WK_NOCONS"),
    ("walker only (no generator, no consumer)", "-DMDPP_ABL_WK_NOCONS -DMDPP_ABL_WK_NOGEN"),
    ("generator only", "-DMDPP_ABL_WK_NOCONS -DMDPP_ABL_WK_NOWALK"),
    ("consumer only (no wait, others exit)", "-DMDPP_ABL_WK_NOWAIT -DMDPP_ABL_WK_NOGEN -DMDPP_ABL_WK_NOWALK"),
    ("every word accepted (no wedge / tail pass)", "-DMDPP_ABL_WK_NOSLOW"),
    ("park 4", "-DMDPP_WK_PARK=4"),
    ("park 16", "-DMDPP_WK_PARK=16"),
    ("attempts 4", "-DMDPP_WK_ATTEMPTS=4"),
    ("gen batch 8", "-DMDPP_WK_GEN_BATCH=8"),
    ("prio gen 3 walker 2 consumer 1", "-DMDPP_WK_GEN_PRIO=3 -DMDPP_WK_WALKER_PRIO=2 -DMDPP_WK_CONSUMER_PRIO=1"),
    ("prio gen 0 walker 3 consumer 1", "-DMDPP_WK_GEN_PRIO=0 -DMDPP_WK_WALKER_PRIO=3 -DMDPP_WK_CONSUMER_PRIO=1"),
    ("prio all 0", "-DMDPP_WK_GEN_PRIO=0 -DMDPP_WK_WALKER_PRIO=0 -DMDPP_WK_CONSUMER_PRIO=0"),
    ("park 1", "-DMDPP_WK_PARK=1"),
    ("park 1, attempts 6", "-DMDPP_WK_PARK=1 -DMDPP_WK_ATTEMPTS=6"),
    ("park 1, attempts 10", "-DMDPP_WK_PARK=1 -DMDPP_WK_ATTEMPTS=10"),
    ("park 1, prio gen 1 walker 3 consumer 2", "-DMDPP_WK_PARK=1 -DMDPP_WK_GEN_PRIO=1 -DMDPP_WK_WALKER_PRIO=3 -DMDPP_WK_CONSUMER_PRIO=2"),
    ("park 1, prio gen 0 walker 2 consumer 3", "-DMDPP_WK_PARK=1 -DMDPP_WK_GEN_PRIO=0 -DMDPP_WK_WALKER_PRIO=2 -DMDPP_WK_CONSUMER_PRIO=3"),
    ("park 1, prio all 0", "-DMDPP_WK_PARK=1 -DMDPP_WK_GEN_PRIO=0 -DMDPP_WK_WALKER_PRIO=0 -DMDPP_WK_CONSUMER_PRIO=0"),
    ("park 1, gen batch 8", "-DMDPP_WK_PARK=1 -DMDPP_WK_GEN_BATCH=8"),
    ("park 1, gen batch 2", "-DMDPP_WK_PARK=1 -DMDPP_WK_GEN_BATCH=2"),
    ("attempts 10", "-DMDPP_WK_ATTEMPTS=10"),
    ("attempts 12", "-DMDPP_WK_ATTEMPTS=12"),
    ("attempts 16", "-DMDPP_WK_ATTEMPTS=16"),
    ("attempts 12, gen batch 8", "-DMDPP_WK_ATTEMPTS=12 -DMDPP_WK_GEN_BATCH=8"),
    ("attempts 6", "-DMDPP_WK_ATTEMPTS=6"),
    ("attempts 5", "-DMDPP_WK_ATTEMPTS=5"),
    ("attempts 7", "-DMDPP_WK_ATTEMPTS=7"),
    ("attempts 4 (park 1)", "-DMDPP_WK_ATTEMPTS=4"),
    ("attempts 6, gen batch 2", "-DMDPP_WK_ATTEMPTS=6 -DMDPP_WK_GEN_BATCH=2"),
    ("attempts 6, gen batch 8", "-DMDPP_WK_ATTEMPTS=6 -DMDPP_WK_GEN_BATCH=8"),
    ("park 2", "-DMDPP_WK_PARK=2"),
    ("park 6", "-DMDPP_WK_PARK=6"),
    ("park 12", "-DMDPP_WK_PARK=12"),
    ("attempts 12", "-DMDPP_WK_ATTEMPTS=12"),
    ("prio gen 2 walker 3 consumer 1", "-DMDPP_WK_GEN_PRIO=2 -DMDPP_WK_WALKER_PRIO=3 -DMDPP_WK_CONSUMER_PRIO=1"),
    ("prio gen 1 walker 3 consumer 2", "-DMDPP_WK_GEN_PRIO=1 -DMDPP_WK_WALKER_PRIO=3 -DMDPP_WK_CONSUMER_PRIO=2"),
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DMDPP_CF_SHAPES_MIN"]
sys.path.insert(0, ROOT)
from mdp_playground_amd import build as B  # noqa: E402
OBJS = [os.path.splitext(f)[0] + ".o" for f in B.SOURCES if f != "mdpp_continuous_fast.hip"]


def build(sel=None):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)

    def one(n):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + VARIANTS[n][1].split() +
                              ["-c", os.path.join(CSRC, "mdpp_continuous_fast.hip"), "-o", os.path.join(OUT, f"cf_{n}.o")],
                              stdout=subprocess.DEVNULL)
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(one, [n for n in range(len(VARIANTS)) if sel is None or n in sel]))


def run(sel=None, rng="numpy"):
    for n, (name, v) in enumerate(VARIANTS):
        obj = os.path.join(OUT, f"cf_{n}.o")
        if not os.path.exists(obj) or (sel is not None and n not in sel):
            continue
        so = os.path.join("/tmp", f"libmdpp_w{n}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", so] + [os.path.join(CSRC, o) for o in OBJS] + [obj])
        code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from mdp_playground_amd import _capi
_capi.LIB_PATH = {so!r}
from mdp_playground_amd import RLToyVectorEnv
import bench
wl = bench.WORKLOADS["cfg5"]
N, F = 65536, 512
env = RLToyVectorEnv(num_envs=N, autoreset="same_step", rng={rng!r}, **wl["config"])
acts = bench.action_rotation(wl, F, N, env.device, 12345)
out = env.alloc_rollout(F)
for j in range(2): env.rollout(acts[j], out)
torch.cuda.synchronize()
us = []
for r in range(3):
    env.timer_begin()
    for j in range(5): env.rollout(acts[j % len(acts)], out)
    us.append(env.timer_end() * 1e3 / 5)
print("%-46s %-60s %8.1f us/launch (min %7.1f)  %.3f of 8 TB/s" % ({name!r}, {v!r}, sorted(us)[1], min(us), 102*N*F/(sorted(us)[1]*1e-6)/8e12), flush=True)
"""
        subprocess.run([sys.executable, "-c", code], timeout=300)
        os.remove(so)


if __name__ == "__main__":
    args = sys.argv[1:]
    sel = None
    if args and args[0] == "build":
        build([int(x) for x in args[1:]] or None)
    else:
        run([int(x) for x in args] or None)
