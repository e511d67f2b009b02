"""The post-processor's oracle (oracle/mdpp_oracle.c ora_p_*) vs the reference's GymEnvWrapper.

tests/golden/w_*.npz were produced by tools/refgen/gen_golden_wrapper.py running
mdp_playground/envs/gym_env_wrapper.py around a deterministic fake env.  Fed the same inner-env
outputs and seeded with the wrapper's post-construction generator state, the oracle must reproduce the
noisy actions, observations (every pixel / float bit), float64 reward bit patterns and the generator's
end state."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as ora

with open(os.path.join(gu.GOLDEN, "wrapper_cases.json")) as _f:
    WCASES = json.load(_f)


def make_post_oracle(case, g):
    cfg, kind = case["config"], case["kind"]
    kw = dict(state_space_type=cfg["state_space_type"], delay=cfg.get("delay", 0),
              transition_noise=cfg.get("transition_noise"), reward_noise=cfg.get("reward_noise"),
              reward_scale=cfg.get("reward_scale", 1.0), reward_shift=cfg.get("reward_shift", 0.0),
              term_state_reward=cfg.get("term_state_reward", 0.0))
    if kind == "continuous":
        kw.update(obs_dim=g["base_obs"].shape[-1], obs_dtype=g["base_obs"].dtype)
    else:
        kw.update(n_actions=6)
    if kind == "image":
        kw.update(image_shape=g["base_obs"].shape[2:], image_transforms=cfg["image_transforms"],
                  image_padding=cfg.get("image_padding", 20), image_sh_quant=cfg.get("image_sh_quant", 1))
    return ora.PostOracle(**kw)


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name", sorted(WCASES))
def test_post_oracle_reproduces_gym_env_wrapper(name):
    case, g = WCASES[name], gu.load(name)
    E, T = g["base_reward"].shape
    for e in range(E):
        o = make_post_oracle(case, g)
        o.set_rng(g["rng0"][e])
        ob0 = o.reset(g["init_base_obs"][e])
        assert same_bits(ob0, g["init_obs"][e]), (name, e)
        for t in range(T):
            if case["kind"] != "continuous":
                assert o.action(int(g["action"][e, t])) == int(g["action_env"][e, t]), (name, e, t)
            ob, r = o.step(g["base_obs"][e, t], g["base_reward"][e, t], g["base_done"][e, t])
            assert same_bits(ob, g["obs"][e, t]), (name, e, t)
            assert np.float64(r).view(np.uint64) == g["reward"][e, t].view(np.uint64), (name, e, t, r, g["reward"][e, t])
            if g["reset_after"][e, t]:
                assert same_bits(o.reset(g["reset_base_obs"][e, t]), g["reset_obs"][e, t]), (name, e, t)
        assert np.array_equal(o.get_rng(), g["rng_end"][e]), (name, e)


def test_pairwise_sum_is_numpy_sum():
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 7, 8, 9, 16, 17, 31, 64, 127, 128):
        for _ in range(50):
            x = np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.integers(-8, 8, size=n))
            assert ora.lib().ora_np_pairwise_sum(x.ctypes.data, n) == (np.sum(x) if n else 0.0)
