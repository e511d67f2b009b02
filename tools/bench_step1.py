"""One launch = one step (mdpp_step): the step1 kernels against the rollout kernels with K = 1.

    python tools/bench_step1.py [cfg2 cfg3 cfg5 ...] [--check] [--variants]

Per workload of bench.py's WORKLOADS: (a) --check: a handle on the default dispatch against a handle with NO_STEP1 on
the same actions, every output of every step and the end state / stream states bit for bit, single steps interleaved with
fused rollouts (the start-state queue is shared); (b) timing: HIP events over 500 eager steps, the host's enqueue time per
call, a replayed graph of 64 steps.  --variants: the tuning knobs of mdpp_discrete_step1.hip (environment variables read at
table upload).  Writes one JSON line per measurement.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402
from mdp_playground_amd import RLToyVectorEnv             # noqa: E402


def make(wl, N, rng="numpy", opts=()):
    env = RLToyVectorEnv(num_envs=N, autoreset="same_step", rng=rng, **wl["config"])
    if opts:
        env.set_kernel_options(*opts)
    env.reset()
    return env


def actions(wl, env, T, N, seed=0):
    g = np.random.default_rng(seed)
    if wl["kind"] == "discrete":
        return torch.as_tensor(g.integers(0, 8, size=(T, N)).astype(np.int32), device=env.device)
    D = env.mdps[0].D
    return torch.as_tensor(g.uniform(-1, 1, size=(T, N, D)).astype(np.float32), device=env.device)


def eq(x, y):
    if x.dtype.is_floating_point:
        return torch.equal(x.view(torch.int32), y.view(torch.int32))
    return torch.equal(x, y)


def check(name, wl, N, rng):
    a, b = make(wl, N, rng), make(wl, N, rng, ("NO_STEP1",))
    T = 96
    acts = actions(wl, a, T + 40 + T, N)
    bad = 0
    for phase, (k0, k1) in enumerate([(0, T), (T + 40, T + 40 + T)]):
        for k in range(k0, k1):
            ra, rb = a.step(acts[k]), b.step(acts[k])
            for x, y in zip(ra[:4], rb[:4]):
                if not eq(x, y):
                    bad += 1
        if phase == 0:      # a fused rollout in between: the queue of start states is shared with the rollout kernels
            ra, rb = a.rollout(acts[T:T + 40]), b.rollout(acts[T:T + 40])
            for x, y in zip(ra, rb):
                if not eq(x, y):
                    bad += 1
    sa, sb = a.get_augmented_state(), b.get_augmented_state()
    for k in sa:
        if isinstance(sa[k], np.ndarray) and not np.array_equal(sa[k], sb[k], equal_nan=True):
            bad += 1
    if rng == "numpy":
        for s in (0, 1):
            if not np.array_equal(a.get_rng_streams(s), b.get_rng_streams(s)):
                bad += 1
    print(json.dumps({"check": name, "rng": rng, "N": N, "kernel": a.rollout_kernel_name(1), "against": b.rollout_kernel_name(1),
                      "mismatches": bad, "status_bits": int((a.status() != 0).sum())}), flush=True)
    a.close(); b.close()
    return bad


def timing(name, wl, N, rng, opts=(), label=""):
    env = make(wl, N, rng, opts)
    dev = env.device
    acts = actions(wl, env, 64, N)
    a1 = acts[0].contiguous()
    for _ in range(50):
        env.step(a1)
    torch.cuda.synchronize(dev)
    n1 = 500
    best = None
    for _ in range(5):
        env.timer_begin()
        t0 = time.perf_counter()
        for _ in range(n1):
            env.step(a1)
        t_host = time.perf_counter() - t0
        ms = env.timer_end()
        torch.cuda.synchronize(dev)
        r = (ms * 1e3 / n1, t_host * 1e6 / n1)
        best = r if best is None or r[0] < best[0] else best
    out = {"timing": name, "rng": rng, "label": label, "kernel": env.rollout_kernel_name(1), "eager_us_events": round(best[0], 3),
           "host_enqueue_us": round(best[1], 3)}
    try:
        g = env.step_graph(acts)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(dev)
        gb = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize(dev)
            v = e0.elapsed_time(e1) * 1e3 / (64 * 20)
            gb = v if gb is None or v < gb else gb
        out["graph_us_per_step"] = round(gb, 3)
    except Exception as e:
        out["graph_error"] = repr(e)
    print(json.dumps(out), flush=True)
    env.close()


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")] or ["cfg2"]
    do_check, variants = "--check" in sys.argv, "--variants" in sys.argv
    rngs = ["numpy", "philox"] if "--philox" in sys.argv else ["numpy"]
    bad = 0
    for name in args:
        wl = bench.WORKLOADS[name]
        N = wl["envs"]
        for rng in rngs:
            if do_check:
                bad += check(name, wl, N, rng)
                bad += check(name, wl, 1000, rng)          # a ragged last block
            timing(name, wl, N, rng, ("NO_STEP1",), "rollout kernel, K = 1")
            timing(name, wl, N, rng, (), "default")
            if variants and wl["kind"] == "discrete":
                for wg in ("64", "256"):
                    for rounds, fill in (("1", "6"), ("1", "1"), ("6", "6"), ("0", "6")):
                        os.environ.update(MDPP_STEP1_WG=wg, MDPP_STEP1_ROUNDS=rounds, MDPP_STEP1_FILL=fill)
                        timing(name, wl, N, rng, (), f"WG={wg} rounds={rounds} fill={fill}")
                for k in ("MDPP_STEP1_WG", "MDPP_STEP1_ROUNDS", "MDPP_STEP1_FILL"):
                    os.environ.pop(k, None)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
