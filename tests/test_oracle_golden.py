"""Oracle (oracle/mdpp_oracle.c) vs the reference itself.

Every fixture under tests/golden/ was produced by tools/refgen/gen_golden.py
running the reference RLToyEnv (mdp_playground/envs/rl_toy_env.py) in the build
container.  Here the oracle is seeded with the reference's post-construction
PCG64 states, fed the same actions and reset schedule, and must reproduce obs,
reward (float64 bit pattern) and done for every step, plus the RNG end state.
"""
import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as ora


@pytest.mark.parametrize("name", gu.DISCRETE)
def test_discrete_rollouts_bit_exact(name):
    g = gu.load(name)
    E, T = g["action"].shape
    for e in range(E):
        o = gu.discrete_oracle_from_golden(name, g, e)
        # the env RNG was re-seeded and drew the initial state at the end of
        # __init__ (:831-833): start from a fresh PCG64(seed) and reset()
        seed_env = int(g["seed_dict"][e][0])
        fresh = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed_env)))
        o.set_rng(ora.pcg_words(fresh), g["rng_space"][e])
        s0 = o.reset()
        assert s0 == int(g["init_state"][e])
        assert np.array_equal(o.get_rng()[0], g["rng_env"][e])
        obs, rew, done, ro = o.rollout(g["action"][e], g["reset_after"][e])
        assert np.array_equal(obs, g["obs"][e].astype(np.int64)), name
        assert np.array_equal(done, g["done"][e]), name
        assert np.array_equal(rew.view(np.uint64), g["reward"][e].view(np.uint64)), name
        ra = g["reset_after"][e]
        assert np.array_equal(ro[ra], g["reset_obs"][e][ra].astype(np.int64))


@pytest.mark.parametrize("name", gu.IRRELEVANT)
def test_discrete_irrelevant_features_rollouts_bit_exact(name):
    """irrelevant_features=True: Tuple observations/actions, a second transition table and a
    second P-noise generator (observation_spaces[1]); reset() draws both start states from the env
    generator (rl_toy_env.py:2028-2035, :2063-2092, :2255-2264)."""
    g = gu.load(name)
    E, T, _ = g["action"].shape
    for e in range(E):
        o = gu.discrete_oracle_from_golden(name, g, e)
        seed_env = int(g["seed_dict"][e][0])
        fresh = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed_env)))
        o.set_rng(ora.pcg_words(fresh), g["rng_space"][e])
        o.set_rng_irr(g["rng_space_irr"][e])
        s0 = o.reset()
        assert list(s0) == [int(x) for x in g["init_state"][e]]
        assert np.array_equal(o.get_rng()[0], g["rng_env"][e])
        obs, rew, done, ro = o.rollout(g["action"][e], g["reset_after"][e])
        assert np.array_equal(obs, g["obs"][e].astype(np.int64)), name
        assert np.array_equal(done, g["done"][e]), name
        assert np.array_equal(rew.view(np.uint64), g["reward"][e].view(np.uint64)), name
        ra = g["reset_after"][e]
        assert np.array_equal(ro[ra], g["reset_obs"][e][ra].astype(np.int64))


@pytest.mark.parametrize("name", gu.GRID)
def test_grid_rollouts_bit_exact(name):
    """Grid envs (rl_toy_env.py:1727-1778, :1947-1965): moves with clipping, actions outside the
    action space applied as noops, noisy actions re-drawn from the action space's generator,
    Manhattan-distance / sparse rewards, the latched target flag, reset() from the feature space."""
    g = gu.load(name)
    E, T, G = g["action"].shape
    for e in range(E):
        o = gu.grid_oracle_from_golden(name)
        sd = g["seed_dict"][e]
        fresh = lambda s: ora.pcg_words(np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(s)))))  # noqa: E731
        o.set_rng(fresh(sd[0]), fresh(sd[5]), g["rng_action"][e])
        assert np.array_equal(o.reset(), g["init_state"][e])
        w = o.get_rng()
        assert np.array_equal(w[0], g["rng_env"][e]) and np.array_equal(w[1][:4], g["rng_space"][e][:4])
        obs, rew, done, ro = o.rollout(g["action"][e], g["reset_after"][e])
        assert np.array_equal(obs, g["obs"][e]), name
        assert np.array_equal(done, g["done"][e]), name
        assert np.array_equal(rew.view(np.uint64), g["reward"][e].view(np.uint64)), name
        ra = g["reset_after"][e]
        assert np.array_equal(ro[ra], g["reset_obs"][e][ra])


@pytest.mark.parametrize("name", gu.CONTINUOUS)
def test_continuous_rollouts_bit_exact(name):
    g = gu.load(name)
    E, T, D = g["action"].shape
    for e in range(E):
        o = gu.continuous_oracle_from_golden(name)
        seed_env = int(g["seed_dict"][e][0])
        seed_space = int(g["seed_dict"][e][5])
        fresh = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed_env)))
        fresh_sp = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed_space)))
        o.set_rng(ora.pcg_words(fresh), ora.pcg_words(fresh_sp))
        s0 = o.reset()
        assert np.array_equal(s0, g["init_state"][e])
        r_env, r_sp = o.get_rng()
        assert np.array_equal(r_env, g["rng_env"][e])
        assert np.array_equal(r_sp, g["rng_space"][e])
        ra = g["reset_after"][e]
        for t in range(T):
            obs, r, is32, d = o.step(g["action"][e, t])
            assert np.array_equal(obs.view(np.uint32), g["obs"][e, t].view(np.uint32)), (name, e, t)
            assert np.array_equal(o.derivs().view(np.uint32), g["sd"][e, t].view(np.uint32)), (name, e, t)
            assert d == bool(g["done"][e, t]), (name, e, t)
            assert np.float64(r).view(np.uint64) == g["reward"][e, t].view(np.uint64), \
                (name, e, t, r, g["reward"][e, t])
            if ra[t]:
                assert np.array_equal(o.reset(), g["reset_obs"][e, t])


@pytest.mark.parametrize("name", ["d_stats", "g_stats", "c_stats"])
def test_episode_statistics_match_the_reference(name):
    """The env object's per-episode noise statistics (total_abs_noise_in_reward_episode, total_reward_episode,
    total_noisy_transitions_episode, total_abs_noise_in_transition_episode; logged at every reset() and cleared,
    rl_toy_env.py:2231-2247, :2360-2369) after every step of reference-generated trajectories: float64 bit patterns,
    and at every reset() the figures it logged (the oracle's `last`)."""
    g = gu.load(name)
    E, T = g["action"].shape[:2]
    fresh = lambda s: ora.pcg_words(np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(s)))))  # noqa: E731
    for e in range(E):
        sd = g["seed_dict"][e]
        if name.startswith("d_"):
            o = gu.discrete_oracle_from_golden(name, g, e)
            o.set_rng(fresh(sd[0]), g["rng_space"][e])
        elif name.startswith("g_"):
            o = gu.grid_oracle_from_golden(name)
            o.set_rng(fresh(sd[0]), fresh(sd[5]), g["rng_action"][e])
        else:
            o = gu.continuous_oracle_from_golden(name)
            o.set_rng(fresh(sd[0]), fresh(sd[5]))
        o.reset()
        ra = g["reset_after"][e]
        for t in range(T):
            o.step(g["action"][e, t] if not name.startswith("d_") else int(g["action"][e, t]))
            cur, _ = o.get_stats()
            want = g["stats"][e, t]
            got = np.concatenate([cur[:3], [want[3]], cur[3:]]) if name.startswith("c_") else np.concatenate([cur, [want[3]]])
            if name.startswith("c_"):
                got[2] = want[2]                     # (noisy transitions: not kept for continuous envs; the reference leaves 0)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (name, e, t, got, want)
            if ra[t]:
                o.reset()
                _, last = o.get_stats()
                assert np.array_equal(last[:2], want[:2]) and last[-1] == want[3], (name, e, t)
                assert not np.any(o.get_stats()[0])
