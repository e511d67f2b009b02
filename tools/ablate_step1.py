#!/usr/bin/env python3
"""Macro-knob variants of a ONE-STEP kernel source (mdpp_continuous_step1.hip, mdpp_discrete_step1.hip), compiled in the build
container and timed on the GPU box as single steps: a replayed graph of 64 mdpp_step launches, us per step.

    python3 tools/ablate_step1.py build <source.hip> "NAME:-DFLAG=1 -DOTHER" ...
    python3 tools/ablate_step1.py run   <source.hip> <workload> <rng> NAME ...       (GPU box)

Timing-only builds (MDPP_ABL_*) produce wrong results by design: nothing here checks them."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import tools.ablate as A  # noqa: E402
from mdp_playground_amd import build as B  # noqa: E402


def run(src, wname, rng, names):
    objs = [os.path.join(A.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != src]
    for name in names:
        so = os.path.join(A.OUT, f"libmdpp__{name}.so")
        if name == "shipped":
            so = B.OUT
        else:
            subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so] + objs + [A.obj_of(src, name)])
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); import torch\n"
                f"from mdp_playground_amd import _capi; _capi.LIB_PATH = {so!r}\n"
                "import bench\n"
                f"wl = bench.WORKLOADS[{wname!r}]; N = wl['envs']\n"
                f"env = bench.make_env(wl, N, torch.device('cuda', 0), {rng!r}); env.reset()\n"
                "acts = bench.make_actions(wl, 64, N, env.device, 1)\n"
                "g = env.step_graph(acts)\n"
                "for j in range(3): g.replay()\n"
                "torch.cuda.synchronize(); best = 1e9\n"
                "for rep in range(5):\n"
                "    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)\n"
                "    e0.record()\n"
                "    for j in range(20): g.replay()\n"
                "    e1.record(); torch.cuda.synchronize()\n"
                "    best = min(best, e0.elapsed_time(e1) * 1e3 / (64 * 20))\n"
                f"print({name!r}, env.rollout_kernel_name(1), '%.2f us per step (replayed graph of 64)' % best, flush=True)\n")
        try:
            r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
            print(r.stdout.strip() if r.returncode == 0 else f"{name} FAILED: {r.stderr[-800:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{name} TIMED OUT", flush=True)
        if name != "shipped":
            os.remove(so)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        A.build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
