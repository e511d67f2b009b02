"""Times baseline/py_step.py (the pure-Python restatement of the reference's step()) on the host:
one core, and one process per core (SURVEY.md §8d Leg B).  Called by bench.py BEFORE the process
initialises the GPU runtime (the all-core leg forks)."""
from __future__ import annotations

import os
import platform
import time

import numpy as np

from . import py_step


def _actions(m, rng, n):
    if m.kind == "discrete" and m.irrelevant:
        return [(int(a), int(b)) for a, b in zip(rng.integers(0, m.A, size=n), rng.integers(0, m.A_irr, size=n))]
    if m.kind == "discrete":
        return [int(a) for a in rng.integers(0, m.A, size=n)]
    return list(rng.uniform(-m.action_space_max, m.action_space_max, size=(n, m.D)).astype(np.float32))


def _worker(args):
    config, seconds, wid = args
    from mdp_playground_amd import mdp as mdp_mod
    m = mdp_mod.build_mdp(config)
    env = py_step.from_mdp(m, mdp_mod.new_generator(1000 + wid), mdp_mod.new_generator(2000 + wid),
                           mdp_mod.new_generator(3000 + wid))
    env.reset()
    acts = _actions(m, np.random.default_rng(wid), 4096)
    steps, t0 = 0, time.perf_counter()
    while True:
        for a in acts:                      # the reference's test-style loop: step, reset on done
            _, _, done, _ = env.step(a)
            if done:
                env.reset()
        steps += len(acts)
        el = time.perf_counter() - t0
        if el >= seconds:
            return steps, el


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def measure(config, kind, seconds_one=10.0, seconds_all=3.0, all_cores=True):
    """-> (one-core record, all-core record) in bench.py's cpu_baseline format, or (None, None) when
    this config has no pure-Python baseline (grid, move_along_a_line, image observations)."""
    import contextlib
    import io
    import multiprocessing as mp
    if (kind == "grid" or config.get("image_representations")
            or config.get("reward_function") == "move_along_a_line"):
        return None, None
    with contextlib.redirect_stdout(io.StringIO()):
        steps, el = _worker((config, seconds_one, 0))
    ncpu = os.cpu_count() or 1
    one = {"value": steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port", "form": "python-restatement",
           "sample": f"{steps} env-steps of the same workload, one env object stepped in a Python loop with "
                     f"random actions and reset on done (baseline/py_step.py: the reference's step() restated, "
                     f"validated against the reference-generated goldens; reference : restatement speed ratio in "
                     f"profiles/py_baseline_ratio.json, folded into `reference_equivalent`) on 1 host core in {el:.1f} s; "
                     f"host: {_cpu_model()}, {ncpu} cores",
           "sample_short": f"{steps} env-steps, 1 env, Python step() loop (baseline/py_step.py), 1 core, {el:.1f} s; {_cpu_model()}"}
    if not all_cores:
        return one, None
    with mp.get_context("fork").Pool(ncpu) as pool, contextlib.redirect_stdout(io.StringIO()):
        res = pool.map(_worker, [(config, seconds_all, w) for w in range(ncpu)])
    total, wall = sum(r[0] for r in res), max(r[1] for r in res)
    allc = {"value": total / wall, "unit": "env-steps/s", "cores": ncpu, "kind": "port", "form": "python-restatement",
            "sample": f"{total} env-steps, one baseline/py_step.py process per core (no inter-process traffic) "
                      f"for {wall:.1f} s"}
    return one, allc
