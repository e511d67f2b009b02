"""baseline/py_step.py (the pure-Python CPU baseline bench.py times) vs the reference
itself: every discrete and continuous `move_to_a_point` fixture of tests/golden/ (made by
tools/refgen/gen_golden.py running the reference) must be reproduced step for step — observations,
float64 reward bit patterns, done flags, reset observations."""
import numpy as np
import pytest

import golden_util as gu
from baseline import py_step
from oracle import oracle as ora


def _fresh(seed):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(seed))))


def _rewardable(g, e):
    return {tuple(int(s) for s in k): float(v) for k, v in zip(g[f"rew_keys_{e}"], g[f"rew_vals_{e}"])}


@pytest.mark.parametrize("name", gu.DISCRETE + gu.IRRELEVANT)
def test_py_discrete_matches_reference_goldens(name):
    g = gu.load(name)
    p = gu.discrete_params(gu.CASES[name]["config"])
    irr = "P_irr" in g.files
    for e in range(g["action"].shape[0]):
        env = py_step.PyDiscreteEnv(
            g["P"][e], _rewardable(g, e), g[f"terminal_states_{e}"], g["init_dist"][e], sequence_length=p["L"],
            delay=p["delay"], reward_every_n_steps=p["every_n"], transition_noise=p["transition_noise"],
            reward_noise=p["reward_noise"], reward_scale=p["reward_scale"], reward_shift=p["reward_shift"],
            term_state_reward=p["term_state_reward"], env_rng=_fresh(g["seed_dict"][e][0]), space_rng=g["rng_space"][e],
            P_irr=g["P_irr"][e] if irr else None, init_dist_irr=g["init_dist_irr"][e] if irr else None,
            space_irr_rng=g["rng_space_irr"][e] if irr else None,
            reward_matrix=g["rew_matrix"][e] if "rew_matrix" in g.files else None)
        assert np.array_equal(env.reset(), g["init_state"][e])
        ra = g["reset_after"][e]
        for t in range(g["action"].shape[1]):
            a = g["action"][e, t]
            obs, r, done, _ = env.step(tuple(int(x) for x in a) if irr else int(a))
            assert np.array_equal(obs, g["obs"][e, t]), (name, e, t)
            assert np.float64(r).view(np.uint64) == g["reward"][e, t].view(np.uint64), (name, e, t, r)
            assert bool(done) == bool(g["done"][e, t]), (name, e, t)
            if ra[t]:
                assert np.array_equal(env.reset(), g["reset_obs"][e, t]), (name, e, t)


@pytest.mark.parametrize("name", [n for n in gu.CONTINUOUS if "line" not in n])
def test_py_continuous_matches_reference_goldens(name):
    g = gu.load(name)
    p = gu.continuous_params(gu.CASES[name]["config"])
    for e in range(g["action"].shape[0]):
        sd = g["seed_dict"][e]
        env = py_step.PyContinuousEnv(
            p["D"], p["relevant_indices"], order=p["order"], inertia=p["inertia"], time_unit=p["time_unit"],
            state_space_max=p["state_space_max"], action_space_max=p["action_space_max"],
            target_point=p["target_point"] if "target_point" in gu.CASES[name]["config"] else None,
            target_radius=p["target_radius"], make_denser=p["make_denser"],
            action_loss_weight=p["action_loss_weight"], transition_noise=p["transition_noise"],
            reward_noise=p["reward_noise"], delay=p["delay"], reward_every_n_steps=p["every_n"],
            reward_scale=p["reward_scale"], reward_shift=p["reward_shift"], term_state_reward=p["term_state_reward"],
            box_lo=p["box_lo"], box_hi=p["box_hi"], env_rng=_fresh(sd[0]), space_rng=_fresh(sd[5]))
        assert np.array_equal(env.reset(), g["init_state"][e])
        ra = g["reset_after"][e]
        for t in range(g["action"].shape[1]):
            obs, r, done, _ = env.step(g["action"][e, t].copy())
            assert np.array_equal(obs.view(np.uint32), g["obs"][e, t].view(np.uint32)), (name, e, t)
            assert np.float64(r).view(np.uint64) == g["reward"][e, t].view(np.uint64), (name, e, t, r)
            assert bool(done) == bool(g["done"][e, t]), (name, e, t)
            if ra[t]:
                assert np.array_equal(env.reset(), g["reset_obs"][e, t]), (name, e, t)


def test_py_baseline_from_mdp_equals_oracle_cfg2():
    """from_mdp() (what bench.py uses) on BASELINE cfg 2 against the C oracle over 3 000 steps with resets."""
    from mdp_playground_amd import mdp as mdp_mod
    import bench
    m = mdp_mod.build_mdp(bench.WORKLOADS["cfg2"]["config"])
    env = py_step.from_mdp(m, mdp_mod.new_generator(5), mdp_mod.new_generator(6))
    o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P, m.reward_table(),
                           m.terminal_states, m.init_dist, m.transition_noise, m.reward_noise, m.reward_scale,
                           m.reward_shift, m.term_state_reward)
    o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator(5)), mdp_mod.pcg64_words(mdp_mod.new_generator(6)))
    assert int(env.reset()) == o.reset()
    acts = np.random.default_rng(0).integers(0, m.A, size=3000).astype(np.int32)
    eo, er, ed, ero = o.rollout(acts, None)          # reset on done
    for t, a in enumerate(acts):
        obs, r, done, _ = env.step(int(a))
        assert int(obs) == eo[t] and r == er[t] and bool(done) == bool(ed[t]), t
        if done:
            assert int(env.reset()) == ero[t]
