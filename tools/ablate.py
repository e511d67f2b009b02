#!/usr/bin/env python3
"""Macro-knob variants of ONE kernel source, compiled in the build container and timed on the GPU box.

    python3 tools/ablate.py build <source.hip> "NAME:-DFLAG=1 -DOTHER" ...      (here: hipcc cross-compiles, in parallel)
    python3 tools/ablate.py run   <source.hip> [workload] [rng] NAME ...         (GPU box: link + time, HIP events;
                                                                                 key=value: config overrides, None drops a key)

Objects live in build/ablate/ (git-ignored, but they travel with the gpurun snapshot).  `run` times 3 x 20 fused
launches of the bench workload with the bench's rotating action tensors and prints the best average per launch."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "mdp_playground_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ablate")
sys.path.insert(0, ROOT)
from mdp_playground_amd import build as B  # noqa: E402


def obj_of(src, name):
    return os.path.join(OUT, f"{src.replace('.hip', '')}__{name}.o")


def build(src, variants):
    os.makedirs(OUT, exist_ok=True)

    def one(v):
        name, flags = (v.split(":", 1) + [""])[:2]
        cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + flags.split() + ["-c", os.path.join(CSRC, src), "-o", obj_of(src, name)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return name, r.returncode, r.stdout[-2000:]
    with ThreadPoolExecutor(max_workers=min(7, len(variants))) as ex:
        for name, rc, out in ex.map(one, variants):
            print(name, "ok" if rc == 0 else "FAILED\n" + out, flush=True)


def run(src, wname, rng, names):
    disable = [n.split("=", 1)[1] for n in names if n.startswith("disable=")]           # disable=NO_LEAN,...: kernel options
    over = {n.split("=", 1)[0]: eval(n.split("=", 1)[1]) for n in names if "=" in n and not n.startswith("disable=")}   # key=python-literal
    names = [n for n in names if "=" not in n]
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != src]
    for name in names:
        so = os.path.join(OUT, f"libmdpp__{name}.so")
        if name == "shipped":
            so = B.OUT
        else:
            subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so] + objs + [obj_of(src, name)])
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); import torch\n"
                f"from mdp_playground_amd import _capi; _capi.LIB_PATH = {so!r}\n"
                "from mdp_playground_amd import RLToyVectorEnv; import bench\n"
                f"wl = bench.WORKLOADS[{wname!r}]; N = wl['envs']; F = min(512, wl.get('fuse_max', 512))\n"
                f"cfg = dict(wl['config'], **{over!r}); cfg = {{k: v for k, v in cfg.items() if v is not None}}\n"
                f"env = RLToyVectorEnv(num_envs=N, autoreset='same_step', rng={rng!r}, **cfg)\n"
                f"env.set_kernel_options(*{disable!r}[0].split(',')) if {disable!r} else None\n"
                "acts = bench.action_rotation(wl, F, N, env.device, 12345); out = env.alloc_rollout(F)\n"
                "for j in range(5): env.rollout(acts[j % len(acts)], out)\n"
                "torch.cuda.synchronize(); best = 1e9\n"
                "for rep in range(3):\n"
                "    env.timer_begin()\n"
                "    for j in range(20): env.rollout(acts[j % len(acts)], out)\n"
                "    best = min(best, env.timer_end() / 20)\n"
                f"print({name!r}, env.rollout_kernel_name(F), '%.1f us per launch' % (best * 1e3), "
                "'%.3f of 8 TB/s' % (wl['alg_bytes_fused'] * N * F / (best * 1e-3) / 8e12), flush=True)\n")
        try:                # (a variant that removes a producer role can spin to its bound on every step: give up after 2 minutes)
            r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            print(r.stdout.strip() if r.returncode == 0 else f"{name} FAILED: {r.stderr[-800:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{name} TIMED OUT after 120 s", flush=True)
        if name != "shipped":
            os.remove(so)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        rest = sys.argv[3:]
        import bench
        wname = rest.pop(0) if rest and rest[0] in bench.WORKLOADS else "cfg2"
        rng = rest.pop(0) if rest and rest[0] in ("numpy", "philox") else "numpy"
        run(sys.argv[2], wname, rng, rest)
