#!/usr/bin/env python3
"""Leg A of the CPU baseline — THIS CONTAINER ONLY (needs /root/reference and the
gymnasium API stand-in, like gen_golden.py; nothing here travels to the GPU box).

Times, on ONE core of this container and in the same process:
  * the ACTUAL reference `RLToyEnv.step()` in the reference tests' loop (construct, step with
    random actions, reset on done; log_level=CRITICAL), and
  * baseline/py_step.py, the pure-Python restatement bench.py times on the GPU box,
for the BASELINE configs, and writes the speed ratio r = reference / restatement to
profiles/py_baseline_ratio.json (bench.py folds it into `cpu_baseline.reference_equivalent`; the round-2 measurement is
profiles/archive/r02_py_baseline_ratio.json).  On the GPU box the reference's own rate is then
(restatement rate measured there) x r.

    python tools/refgen/bench_reference.py [seconds per leg, default 8]
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402

import gen_golden  # noqa: E402  (make_env: the reference under the stand-in)
import bench  # noqa: E402
from baseline import bench_py, py_step  # noqa: E402
from mdp_playground_amd import mdp as mdp_mod  # noqa: E402


def time_loop(env, acts, seconds):
    steps, t0 = 0, time.perf_counter()
    while True:
        for a in acts:
            out = env.step(a)
            if out[2]:
                env.reset()
        steps += len(acts)
        el = time.perf_counter() - t0
        if el >= seconds:
            return steps / el


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    out = {"what": "env-steps/s of the reference RLToyEnv.step() and of baseline/py_step.py, same loop, same "
                   "process, 1 core of the build container; ratio = reference / restatement",
           "command": "python tools/refgen/bench_reference.py", "seconds_per_leg": seconds,
           "host": bench_py._cpu_model(), "numpy": np.__version__, "workloads": {}}
    for w in ("cfg2", "cfg2_noise", "cfg3", "cfg5"):
        wl = bench.WORKLOADS[w]
        m = mdp_mod.build_mdp(wl["config"])
        acts = bench_py._actions(m, np.random.default_rng(0), 4096)
        ref = gen_golden.make_env(wl["config"])
        racts = acts if m.kind == "discrete" else [np.array(a) for a in acts]
        r_ref = time_loop(ref, racts, seconds)
        mine = py_step.from_mdp(m, mdp_mod.new_generator(1000), mdp_mod.new_generator(2000))
        mine.reset()
        r_py = time_loop(mine, acts, seconds)
        out["workloads"][w] = {"reference_steps_per_s": r_ref, "restatement_steps_per_s": r_py,
                               "ratio_reference_over_restatement": r_ref / r_py}
        print(w, out["workloads"][w], flush=True)
    with open(os.path.join(ROOT, "profiles", "py_baseline_ratio.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
