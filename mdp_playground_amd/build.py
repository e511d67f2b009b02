"""Build libmdpp_hip.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m mdp_playground_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the float32/float64
arithmetic of the continuous kernel must not be fused into FMAs (parity with numpy).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libmdpp_hip.so")
SOURCES = ["mdpp_capi.hip", "mdpp_discrete.hip", "mdpp_discrete_wide.hip", "mdpp_discrete_long.hip", "mdpp_discrete_fast.hip", "mdpp_discrete_step1.hip", "mdpp_discrete_pipe.hip", "mdpp_discrete_lean.hip", "mdpp_discrete_lean_next.hip", "mdpp_discrete_lean_noise.hip", "mdpp_discrete_lean_npnoise.hip",
           "mdpp_discrete_quiet.hip", "mdpp_discrete_quiet_nu.hip",
           "mdpp_continuous.hip", "mdpp_continuous_line8.hip",
           "mdpp_continuous_fast.hip", "mdpp_continuous_step1.hip", "mdpp_continuous_line.hip", "mdpp_image.hip", "mdpp_grid.hip", "mdpp_imagec.hip", "mdpp_post.hip", "mdpp_peer.hip"]
HEADERS = ["mdpp_internal.hpp", "mdpp_rng.hpp", "mdpp_pcg64_limbs.inc", "np_ziggurat_tables.inc",
           os.path.join("..", "..", "include", "mdpp.h")]
INCLUDED_SOURCES = {"mdpp_discrete_wide.hip": ["mdpp_discrete.hip"], "mdpp_discrete_long.hip": ["mdpp_discrete.hip"], "mdpp_discrete_lean_next.hip": ["mdpp_discrete_lean.hip"], "mdpp_discrete_lean_noise.hip": ["mdpp_discrete_lean.hip"], "mdpp_discrete_lean_npnoise.hip": ["mdpp_discrete_lean.hip"],
                    "mdpp_continuous_line8.hip": ["mdpp_continuous.hip"], "mdpp_continuous_step1.hip": ["mdpp_continuous_fast.hip"], "mdpp_discrete_quiet_nu.hip": ["mdpp_discrete_quiet.hip"]}   # a .hip that #includes another one
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# The GENERAL kernels keep whole state vectors in registers and spill (k_continuous_step<DMAX=32, OMAX=4>: 3 KB of scratch per lane).
# With the compiler's default, SGPRs are spilled into the lanes of a VGPR, and where that VGPR is itself spilled inside divergent control
# flow the values parked in the inactive lanes are lost: k_continuous_step<32, 4, PHILOX> ended episodes that had not ended (lanes 43-60
# of a wave, whenever another lane of the wave ran the in-step reset; round 6, found by the random configurations against the oracle,
# tools/repro_c14.py).  These translation units spill SGPRs to memory instead; the hand-tuned rollout kernels do not spill and keep
# the default.
SPILL_SAFE = ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]
# (+ the quiet kernel's two units: their Philox instantiations have 32 B of VGPR scratch and the flag costs them 0.4-1.5 %)
EXTRA_FLAGS = {s: SPILL_SAFE for s in ("mdpp_continuous.hip", "mdpp_continuous_line8.hip", "mdpp_discrete.hip", "mdpp_discrete_wide.hip",
                                       "mdpp_discrete_long.hip", "mdpp_discrete_quiet.hip", "mdpp_discrete_quiet_nu.hip")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = [os.path.join(CSRC, d) for d in INCLUDED_SOURCES.get(s, [])]
        if force or _stale(obj, [src] + extra + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    if jobs:
        # the long compiles first (template-heavy kernels: 1.5-3 min each), as many at once as there are cores to spare
        heavy = ("mdpp_continuous_fast", "mdpp_discrete_lean", "mdpp_discrete_quiet", "mdpp_grid", "mdpp_continuous.", "mdpp_continuous_step1")
        jobs.sort(key=lambda c: next((k for k, h in enumerate(heavy) if h in c[-3]), len(heavy)))
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 4) - 1, 7))) as ex:
            for out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(out)
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
