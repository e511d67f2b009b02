#!/usr/bin/env python3
"""Build variants of k_discrete_rollout_pipe / _lean (macro knobs) and time them on the bench workload (GPU box).
usage: python3 tools/ablate_pipe.py [--file mdpp_discrete_lean.hip] [--workload cfg4] "NAME:-DMDPP_PIPE_CHUNK=16 -DMDPP_PIPE_DEPTH=64" ..."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
sys.path.insert(0, ROOT)
from mdp_playground_amd import build as B  # noqa: E402


def main():
    src = "mdpp_discrete_pipe.hip"
    wname, alg = "cfg2", 18
    while len(sys.argv) > 2 and sys.argv[1] in ("--file", "--workload"):
        if sys.argv[1] == "--file":
            src = sys.argv[2]
        else:
            wname = sys.argv[2]
        del sys.argv[1:3]
    outdir = os.path.join(ROOT, "gpurun_out", "ablate_pipe")
    os.makedirs(outdir, exist_ok=True)
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != src]
    variants = [("base", "")] + [tuple(v.split(":", 1)) for v in sys.argv[1:]]
    for name, flags in variants:
        obj = os.path.join(outdir, f"pipe_{name}.o")
        so = os.path.join(outdir, f"libmdpp_{name}.so")
        subprocess.check_call([B._hipcc()] + B.FLAGS + flags.split() + ["-c", os.path.join(CSRC, src), "-o", obj])
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so] + objs + [obj])
        rng = os.environ.get("ABL_RNG", "numpy")
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); import torch\n"
                f"from mdp_playground_amd import _capi; _capi.LIB_PATH = {so!r}\n"
                "from mdp_playground_amd import RLToyVectorEnv; import bench\n"
                f"wl = bench.WORKLOADS[{wname!r}]; N = wl['envs']; F = min(512, wl.get('fuse_max', 512))\n"
                f"env = RLToyVectorEnv(num_envs=N, autoreset='same_step', rng={rng!r}, **wl['config'])\n"
                "acts = bench.action_rotation(wl, F, N, env.device, 12345); out = env.alloc_rollout(F)\n"
                "for j in range(5): env.rollout(acts[j % len(acts)], out)\n"
                "torch.cuda.synchronize(); best = 1e9\n"
                "for rep in range(3):\n"
                "    env.timer_begin()\n"
                "    for j in range(20): env.rollout(acts[j % len(acts)], out)\n"
                "    best = min(best, env.timer_end() / 20)\n"
                f"print({name!r}, env.rollout_kernel_name(F), '%.1f us per launch' % (best * 1e3), '%.3f of 8 TB/s' % (wl['alg_bytes_fused'] * N * F / (best * 1e-3) / 8e12))\n")
        subprocess.check_call([sys.executable, "-c", code])


if __name__ == "__main__":
    main()
