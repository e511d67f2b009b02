#!/usr/bin/env python3
"""Golden vectors for the batched post-processor made by running the
reference's GymEnvWrapper (mdp_playground/envs/gym_env_wrapper.py) around a deterministic fake env.

THIS CONTAINER ONLY, like gen_golden.py: needs /root/reference and the gymnasium API stand-in.
gym_env_wrapper.py hard-imports `ale_py` and `gymnasium.wrappers.AtariPreprocessing` (:6, :15-16),
neither of which its step()/reset()/get_transformed_image() arithmetic touches: an empty `ale_py`
module and a dummy AtariPreprocessing class are put in place before the import.

What is recorded per case ([E, T, ...], E wrapper instances seeded seed0 + e):
  * the wrapper config, the PCG64 state of the wrapper's generator right after construction
  * per step: the agent's action, the action the inner env received (discrete action noise),
    the inner env's (obs, reward, done), and what the wrapper returned (obs, reward)
  * whether reset() was called after the step, and the observation reset() returned
One upstream defect is worked around and recorded in VERSIONS.txt: the `done` branch (:403-410)
evaluates `self.reward_buffer * self.reward_scale + self.reward_shift` on a Python LIST, which raises
TypeError for every config (list * float, or list + float); the evident intent is numpy arithmetic,
so after every reset() the generator wraps the list in a list subclass whose `*` is numpy's.  Steps
that are not `done` run the reference untouched.
"""
import contextlib
import io
import json
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import mdp_playground.envs  # noqa: E402,F401  (its own GymEnvWrapper import fails inside and is caught upstream)
sys.modules["ale_py"] = types.ModuleType("ale_py")
import gymnasium  # noqa: E402


class _AtariPreprocessing:      # never instantiated (atari_preprocessing is not in any config here)
    pass


gymnasium.wrappers.AtariPreprocessing = _AtariPreprocessing
from gymnasium.spaces import Box, Discrete  # noqa: E402
from mdp_playground.envs.gym_env_wrapper import GymEnvWrapper  # noqa: E402

from gen_golden import pcg_state  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden"))


class ArrList(list):
    """reward_buffer with numpy's `*` (see the module docstring)."""

    def __mul__(self, k):
        return np.asarray(self, dtype=np.float64) * k


class FakeEnv(gymnasium.Env):
    """Deterministic inner env: observation, reward and done are arithmetic functions of (instance,
    step, action) with dyadic rewards, so every sum the wrapper forms is exact."""

    def __init__(self, kind, e, n_actions=6, obs_shape=(5,), obs_dtype=np.float32, done_every=13):
        self.kind, self.e, self.done_every = kind, e, done_every
        self.obs_shape, self.obs_dtype = tuple(obs_shape), np.dtype(obs_dtype)
        if kind == "discrete":
            self.action_space = Discrete(n_actions)
            self.observation_space = Discrete(97)
        elif kind == "image":
            self.action_space = Discrete(n_actions)
            self.observation_space = Box(0, 255, shape=self.obs_shape, dtype=np.uint8)
        else:
            self.action_space = Box(-1.0, 1.0, shape=(3,), dtype=np.float32)
            self.observation_space = Box(-100.0, 100.0, shape=self.obs_shape, dtype=self.obs_dtype)
        self.t = 0
        self.episode = 0

    def _obs(self, a):
        k = self.t * 31 + self.e * 7 + self.episode * 3
        if self.kind == "discrete":
            return (k + int(a)) % 97
        if self.kind == "image":
            idx = np.arange(int(np.prod(self.obs_shape)), dtype=np.int64).reshape(self.obs_shape)
            return ((idx * 7 + k * 13 + int(a) * 29) % 251).astype(np.uint8)
        idx = np.arange(self.obs_shape[0], dtype=np.float64)
        return (np.sin(0.37 * k + idx) * 3.0 + float(np.sum(a))).astype(self.obs_dtype)

    def reset(self, seed=None, options=None):
        self.t = 0
        self.episode += 1
        return self._obs(0 if self.kind != "continuous" else np.zeros(3, np.float32)), {}

    def step(self, action):
        self.t += 1
        a = action if self.kind == "continuous" else int(action)
        sa = float(np.sum(action)) if self.kind == "continuous" else float(a)
        reward = ((self.t * 5 + self.e * 3 + int(round(sa * 4))) % 17 - 6) / 8.0      # dyadic
        done = (self.t % self.done_every) == 0
        return self._obs(a), reward, done, False, {}


CASES = {
    "w_disc_all": dict(kind="discrete", T=90, seeds=8, done_every=13, env={},
                       config=dict(state_space_type="discrete", delay=3, transition_noise=0.25, reward_noise=0.5,
                                   reward_scale=2.5, reward_shift=-1.0, term_state_reward=4.0)),
    "w_disc_plain": dict(kind="discrete", T=40, seeds=4, done_every=11, env={},
                         config=dict(state_space_type="discrete", delay=1)),
    "w_cont_noise": dict(kind="continuous", T=90, seeds=8, done_every=17, env=dict(obs_shape=(5,), obs_dtype="float32"),
                         config=dict(state_space_type="continuous", delay=2, transition_noise=0.125, reward_noise=0.25,
                                     reward_scale=0.5, reward_shift=0.75, term_state_reward=-2.0)),
    "w_cont_f64": dict(kind="continuous", T=60, seeds=4, done_every=9, env=dict(obs_shape=(3,), obs_dtype="float64"),
                       config=dict(state_space_type="continuous", transition_noise=0.5, reward_scale=3.0)),
    "w_img_shift": dict(kind="image", T=50, seeds=6, done_every=12, env=dict(obs_shape=(16, 16, 3)),
                        config=dict(state_space_type="discrete", delay=1, reward_noise=0.125, image_transforms="shift",
                                    image_padding=6, image_sh_quant=2, transition_noise=0.1)),
    "w_img_centre": dict(kind="image", T=20, seeds=3, done_every=7, env=dict(obs_shape=(12, 12, 3)),
                         config=dict(state_space_type="discrete", image_transforms="flip", image_padding=4)),
}


def run_case(name, case):
    E, T = case["seeds"], case["T"]
    rec = {k: [] for k in ("rng0", "action", "action_env", "base_obs", "base_reward", "base_done", "obs", "reward",
                           "reset_after", "reset_base_obs", "reset_obs", "init_base_obs", "init_obs", "rng_end")}
    for e in range(E):
        envkw = dict(case["env"])
        if "obs_dtype" in envkw:
            envkw["obs_dtype"] = np.dtype(envkw["obs_dtype"])
        inner = FakeEnv(case["kind"], e, done_every=case["done_every"], **envkw)
        got = {}
        orig_step = inner.step

        def spy_step(action, _o=orig_step, _g=got):
            out = _o(action)
            _g["action_env"], _g["base"] = action, out
            return out
        inner.step = spy_step
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w = GymEnvWrapper(inner, **dict(case["config"], seed=1000 + e))
        rec["rng0"].append(pcg_state(w._np_random))
        base0 = None
        orig_reset = inner.reset

        def spy_reset(seed=None, options=None, _o=orig_reset, _g=got):
            out = _o(seed=seed)
            _g["reset_base"] = out[0]
            return out
        inner.reset = spy_reset
        obs0, _ = w.reset()
        w.reward_buffer = ArrList(w.reward_buffer)
        rec["init_base_obs"].append(np.asarray(got["reset_base"]))
        rec["init_obs"].append(np.asarray(obs0))
        arng = np.random.default_rng(77 + e)
        row = {k: [] for k in ("action", "action_env", "base_obs", "base_reward", "base_done", "obs", "reward",
                               "reset_after", "reset_base_obs", "reset_obs")}
        for t in range(T):
            if case["kind"] == "continuous":
                a = arng.uniform(-1, 1, size=3).astype(np.float32)
            else:
                a = int(arng.integers(0, inner.action_space.n))
            obs, r, done, trunc, _ = w.step(a)
            bo, br, bd, _, _ = got["base"]
            row["action"].append(a)
            row["action_env"].append(got["action_env"])
            row["base_obs"].append(np.asarray(bo)); row["base_reward"].append(float(br)); row["base_done"].append(bool(bd))
            row["obs"].append(np.asarray(obs)); row["reward"].append(float(r))
            assert bool(done) == bool(bd) and trunc is False
            if done:
                ro, _ = w.reset()
                w.reward_buffer = ArrList(w.reward_buffer)
                row["reset_after"].append(True)
                row["reset_base_obs"].append(np.asarray(got["reset_base"])); row["reset_obs"].append(np.asarray(ro))
            else:
                row["reset_after"].append(False)
                row["reset_base_obs"].append(np.zeros_like(np.asarray(bo))); row["reset_obs"].append(np.zeros_like(np.asarray(obs)))
        for k, v in row.items():
            rec[k].append(np.stack([np.asarray(x) for x in v]))
        rec["rng_end"].append(pcg_state(w._np_random))
    out = {k: np.stack(v) for k, v in rec.items()}
    out["reward"] = out["reward"].astype(np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return {"config": case["config"], "kind": case["kind"], "T": T, "seeds": [1000 + e for e in range(E)],
            "env": case["env"], "done_every": case["done_every"]}


def main():
    meta = {}
    for name, case in CASES.items():
        meta[name] = run_case(name, case)
        print(name, "ok")
    with open(os.path.join(OUT, "wrapper_cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
