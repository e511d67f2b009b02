"""GPU parity: the HIP path (through the C ABI, via RLToyVectorEnv) against
  (1) the golden vectors the reference produced (tests/golden/, bit-exact), and
  (2) the C oracle on freshly seeded inputs at sizes the oracle finishes in seconds.

Bars: discrete obs / done bit-exact, rewards equal to float32(reference float64 reward);
continuous float32 states bit-exact here (north_star allows 1e-6 relative), rewards equal to
float32(reference reward) except where stated.
"""
import warnings

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _venv(**kw):
    from mdp_playground_amd import RLToyVectorEnv
    return RLToyVectorEnv(**kw)


def _seeds_or_cfg(name):
    case = gu.CASES[name]
    cfg = dict(case["config"])
    seeds = case["seeds"]
    if seeds == [None]:
        return dict(num_envs=1, **cfg)
    return dict(seeds=seeds, **cfg)


# ----------------------------------------------------------------------------- discrete
@pytest.mark.parametrize("name", gu.DISCRETE)
def test_discrete_stepwise_vs_reference_golden(name):
    g = gu.load(name)
    E, T = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    obs0 = env._obs
    assert np.array_equal(obs0.cpu().numpy(), g["init_state"].astype(np.int64))
    assert np.array_equal(env.get_rng_streams(0), g["rng_env"])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy(), g["obs"][:, t].astype(np.int64)), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra].astype(np.int64))
    env.close()


@pytest.mark.parametrize("name", [n for n in gu.DISCRETE if gu.CASES[n]["reset"] in ("on_done", "never")])
def test_discrete_fused_rollout_vs_reference_golden(name):
    """One launch for the whole trajectory: same-step autoreset reproduces the reference's
    step(); if done: reset() loop (obs at a terminal step is the next episode's first obs)."""
    g = gu.load(name)
    E, T = g["action"].shape
    mode = gu.CASES[name]["reset"]
    env = _venv(autoreset="same_step" if mode == "on_done" else "disabled", **_seeds_or_cfg(name))
    acts = torch.as_tensor(g["action"].T.astype(np.int32).copy(), device=env.device)
    obs, rew, term, trunc = env.rollout(acts)
    exp_obs = g["obs"].astype(np.int64).copy()
    ra = g["reset_after"]
    exp_obs[ra] = g["reset_obs"].astype(np.int64)[ra]
    assert np.array_equal(obs.cpu().numpy().T, exp_obs)
    assert np.array_equal(term.cpu().numpy().T, g["done"])
    assert np.array_equal(rew.cpu().numpy().T, g["reward"].astype(np.float32))
    assert not trunc.any()
    env.close()


def test_discrete_same_step_autoreset_final_obs():
    name = "d_cfg2"
    g = gu.load(name)
    env = _venv(autoreset="same_step", **_seeds_or_cfg(name))
    for t in range(60):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, info = env.step(a)
        d = g["done"][:, t]
        assert np.array_equal(term.cpu().numpy(), d)
        assert np.array_equal(info["final_obs"].cpu().numpy()[d], g["obs"][:, t][d])
        assert np.array_equal(obs.cpu().numpy()[d], g["reset_obs"][:, t][d])
        assert np.array_equal(obs.cpu().numpy()[~d], g["obs"][:, t][~d])
    env.close()


def _oracle_for(env, i):
    from oracle import oracle as ora
    m = env.mdps[i if env._per_env else 0]
    if m.kind == "grid":
        return ora.GridOracle(m.grid_shape, m.target_point, m.make_denser, m.transition_noise, m.reward_noise,
                              m.reward_every_n_steps, m.reward_scale, m.reward_shift, m.term_state_reward)
    if m.kind == "discrete":
        custom = m.reward_matrix is not None
        o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P,
                               np.zeros(m.S ** m.sequence_length) if custom else m.reward_table(),
                               m.terminal_states, m.init_dist, m.transition_noise,
                               m.reward_noise, m.reward_scale, m.reward_shift, m.term_state_reward)
        if custom:
            o.set_reward_matrix(m.reward_matrix)
        if m.irrelevant:
            o.set_irrelevant(m.P_irr, m.init_dist_irr)
    else:
        o = ora.ContinuousOracle(m.D, m.relevant_indices, m.order, m.inertia, m.time_unit,
                                 m.state_space_max, m.action_space_max, m.target_point,
                                 m.target_radius, m.make_denser, m.action_loss_weight,
                                 m.transition_noise, m.reward_noise, m.delay, m.reward_every_n_steps,
                                 m.reward_scale, m.reward_shift, m.term_state_reward, m.box_lo, m.box_hi)
        if m.reward_function == "move_along_a_line":
            o.set_line_reward(m.sequence_length, m.delay)
        elif m.target_default:
            o.set_target64()
    return o


@pytest.mark.parametrize("noise", [False, True])
def test_discrete_shared_mdp_4096_envs_vs_oracle(noise):
    """BASELINE cfg 2 shape, one shared MDP in LDS, 4096 instances with their own streams;
    every instance is checked against its own oracle instance (same-step autoreset)."""
    cfg = dict(gu.CASES["d_cfg2_noise" if noise else "d_cfg2"]["config"], seed=3)
    N, T = 4096, 96
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 8, size=(T, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    end_env, end_sp = env.get_rng_streams(0), env.get_rng_streams(1)
    for i in range(0, N, 7):
        o = _oracle_for(env, i)
        # the streams exactly as they were uploaded, before the construction-time reset()
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert o.reset() == int(init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i], exp), i
        assert np.array_equal(term[:, i], ed), i
        assert np.array_equal(rew[:, i], er.astype(np.float32)), i
        we, ws = o.get_rng()
        assert np.array_equal(we[:4], end_env[i][:4]) and np.array_equal(ws[:4], end_sp[i][:4])
    env.close()


@pytest.mark.parametrize("name,rng", [("d_custom_pr", "numpy"), ("d_custom_noise", "numpy"),
                                      ("d_custom_noise", "philox")])
def test_discrete_custom_matrices_2048_envs_vs_oracle(name, rng):
    """use_custom_mdp with P and R matrices (reward keyed by the transition's (s, a)): one shared MDP
    in LDS, 2048 instances, fused rollout with same-step autoreset then single steps, every 5th
    instance against its own oracle instance; negative action indices wrap like numpy's."""
    cfg = dict(gu.CASES[name]["config"], seed=11)
    N, T, T1 = 2048, 64, 8
    kw = dict(rng="philox", philox_seed=99) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    A = cfg["action_space_size"]
    acts = np.random.default_rng(6).integers(0, A, size=(T + T1, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts[:T], device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    single = []
    for t in range(T, T + T1):
        o1, r1, d1, _, _ = env.step(torch.as_tensor(acts[t], device=env.device))
        single.append((o1.cpu().numpy().copy(), r1.cpu().numpy().copy(), d1.cpu().numpy().copy()))
    for i in range(0, N, 5):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(99, i)
            assert o.reset() == int(init[i])
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            assert o.reset() == int(init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i], exp[:T]) and np.array_equal(term[:, i], ed[:T]), i
        # (Philox mode: the ziggurat tail's log1p/exp may differ from glibc's in the last ulp)
        same = np.array_equal if rng == "numpy" else (lambda x, y: np.allclose(x, y, rtol=1e-6, atol=1e-6))
        assert same(rew[:, i], er[:T].astype(np.float32)), i
        for k, (o1, r1, d1) in enumerate(single):
            assert o1[i] == exp[T + k] and d1[i] == ed[T + k] and same(r1[i], np.float32(er[T + k])), (i, k)
    env.close()


# ----------------------------------------------------------------------------- quiet rollout kernel
QUIET = {
    "s20_l4_d3": (dict(state_space_size=20, action_space_size=20, sequence_length=4, delay=3,
                       reward_every_n_steps=2, reward_density=0.02, reward_scale=1.5, reward_shift=-0.25,
                       term_state_reward=2.0), {}, 1024),          # full blocks: three roles (E / O / H waves)
    "diam2_s24_l3": (dict(state_space_size=24, action_space_size=12, diameter=2, sequence_length=3, delay=0,
                          terminal_state_density=0.25), {}, 777),
    "irr_8x5": (dict(state_space_size=[8, 5], action_space_size=[8, 5], irrelevant_features=True, delay=2,
                     sequence_length=2, term_state_reward=-1.0), {}, 1030),
    "irr_i32_horizon": (dict(state_space_size=[6, 9], action_space_size=[6, 9], irrelevant_features=True,
                             sequence_length=1, dtype_s=np.int32), dict(max_episode_steps=7), 512),
    "s32_noreset": (dict(state_space_size=32, action_space_size=32, sequence_length=2, delay=1, reward_density=0.05),
                    dict(autoreset="disabled"), 512),           # full blocks, no autoreset: two roles (E / O)
    # the noise variants: P-noise (three roles), reward noise (env stream shared with reset: two roles),
    # both on a ragged batch (one role), reward noise with std 0 (still draws) and a horizon
    "pn_s12": (dict(state_space_size=12, action_space_size=12, sequence_length=2, delay=1, transition_noise=0.25,
                    reward_scale=2.0), {}, 768),
    "rn_s8": (dict(state_space_size=8, action_space_size=8, sequence_length=3, delay=4, reward_noise=0.3,
                   reward_shift=-0.5, term_state_reward=1.0), {}, 512),
    "pn_rn_ragged": (dict(state_space_size=10, action_space_size=10, sequence_length=1, delay=0, transition_noise=0.1,
                          reward_noise=1.5), {}, 700),
    "irr_pn_rn": (dict(state_space_size=[8, 6], action_space_size=[8, 6], irrelevant_features=True, delay=1,
                       sequence_length=2, transition_noise=0.2, reward_noise=0.3), {}, 512),
    "irr_pn_ragged": (dict(state_space_size=[6, 12], action_space_size=[6, 12], irrelevant_features=True, delay=0,
                           sequence_length=1, transition_noise=0.35), {}, 333),
    "pn_rn0_horizon": (dict(state_space_size=20, action_space_size=20, sequence_length=2, delay=2, transition_noise=0.4,
                            reward_noise=0.0, reward_every_n_steps=1), dict(max_episode_steps=9), 256),
    # round 6: the reference's sweep defaults (sequence_length 1, same-step autoreset, no step limit) -- the compile-time form SF
    # -- without a noise key (E / O / H), with reward noise (XR: the third wave evaluates the env stream by position; values
    # used), with the key at sigma 0, a delay line, non-unit rewards; S = 130 is past SF's terminal bit
    "sf_s50": (dict(state_space_size=50, action_space_size=50, sequence_length=1, delay=0, reward_density=0.25,
                    terminal_state_density=0.25), {}, 1024),
    "sf_s50_rn": (dict(state_space_size=50, action_space_size=50, sequence_length=1, delay=2, reward_density=0.25,
                       terminal_state_density=0.25, reward_noise=0.7, reward_scale=1.5, term_state_reward=-1.0), {}, 1024),
    "sf_s24_rn0": (dict(state_space_size=24, action_space_size=24, sequence_length=1, delay=1, reward_density=0.25,
                        terminal_state_density=0.25, reward_noise=0.0), {}, 768),
    "sf_s24_rdist_rn": (dict(state_space_size=24, action_space_size=24, sequence_length=1, delay=1, reward_density=0.25,
                             terminal_state_density=0.25, reward_dist=[0.01, 1], reward_noise=0.2), {}, 512),
    "xr_s20_l2_rn": (dict(state_space_size=20, action_space_size=20, sequence_length=2, delay=1, reward_noise=0.4,
                          reward_every_n_steps=1), dict(max_episode_steps=11), 512),
    "xr_s130_rn_noreset": (dict(state_space_size=130, action_space_size=20, sequence_length=1, delay=0, reward_density=0.1,
                                reward_noise=0.3), dict(autoreset="disabled"), 512),
}


@pytest.mark.parametrize("variant", sorted(QUIET))
def test_discrete_quiet_rollout_kernel_vs_oracle(variant):
    """k_discrete_rollout_quiet (quiet discrete shapes beyond the specialised kernels: larger S and
    L, diameter 2, irrelevant sub-space): every 7th env against its oracle over launches of
    different lengths — short ones go to the general kernel, long ones to the quiet kernel (one role
    for ragged batches, two or three roles on full 256-env blocks), on the same state — incl.
    terminal observations and the env generator's state after every launch (the
    queue of pre-drawn start states must be un-drawn exactly)."""
    from mdp_playground_amd import _capi as capi
    cfg_extra, env_kw, N = QUIET[variant]
    cfg = dict(state_space_type="discrete", action_space_type="discrete", seed=41, **cfg_extra)
    kw = dict(autoreset="same_step")
    kw.update(env_kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = _venv(num_envs=N, **kw, **cfg)
    assert not env.rollout_kernel_name(64).startswith(("k_discrete_rollout_fast", "k_discrete_rollout_pipe"))
    if variant.startswith(("sf_", "xr_")):
        kn = env.rollout_kernel_name(64)
        assert kn.startswith("k_discrete_rollout_quiet<") and "ROLES=3" in kn and ("SF=1" in kn) == variant.startswith("sf_"), kn
    m = env.mdps[0]
    horizon = env_kw.get("max_episode_steps", 0)
    auto = kw["autoreset"] == "same_step"
    r = np.random.default_rng(3)
    init = env._obs.cpu().numpy().copy()
    oracles = []
    for i in range(0, N, 7):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        if m.irrelevant:
            o.set_rng_irr(env.seeded_streams[capi.STREAM_SPACE_IRR][i])
        assert np.array_equal(np.asarray(o.reset()), init[i])
        oracles.append((i, o, [0]))
    for K in (1, 40, 5, 64, 17):
        if m.irrelevant:
            acts = np.stack([r.integers(0, m.A, size=(K, N)), r.integers(0, m.A_irr, size=(K, N))], axis=2).astype(np.int32)
        else:
            acts = r.integers(0, m.A, size=(K, N)).astype(np.int32)
        at = torch.as_tensor(acts, device=env.device)
        if K == 1:
            o1, r1, t1, tr1, info = env.step(at[0])
            obs, rew, term, trunc = (x[None].cpu().numpy() for x in (o1, r1, t1, tr1))
            fin = info["final_obs"].cpu().numpy() if auto else None
        else:
            obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(at))
            fin = None
        env_end = env.get_rng_streams(capi.STREAM_ENV)
        sp_end = env.get_rng_streams(capi.STREAM_SPACE)
        for i, o, ep in oracles:
            for t in range(K):
                st, rr, d = o.step(acts[t, i])
                ep[0] += 1
                tr = bool(horizon) and ep[0] >= horizon
                assert d == bool(term[t, i]) and tr == bool(trunc[t, i]), (variant, K, i, t)
                assert np.float32(rr) == rew[t, i], (variant, K, i, t)
                if auto and (d or tr):
                    if fin is not None:
                        assert np.array_equal(fin[i], np.asarray(st)), (variant, K, i)
                    st = o.reset(explicit=False)
                    ep[0] = 0
                assert np.array_equal(obs[t, i], np.asarray(st)), (variant, K, i, t)
            assert np.array_equal(o.get_rng()[0][:4], env_end[i][:4]), (variant, K, i)
            assert np.array_equal(o.get_rng()[1][:4], sp_end[i][:4]), (variant, K, i)
            if m.irrelevant:
                assert np.array_equal(o.get_rng_irr()[:4], env.get_rng_streams(capi.STREAM_SPACE_IRR)[i][:4]), (variant, K, i)
    assert int(env.status().sum()) == 0          # no bad action, and no bounded wait of the role hand-offs expired
    env.close()


# ----------------------------------------------------------------------------- irrelevant features
@pytest.mark.parametrize("name", gu.IRRELEVANT)
def test_irrelevant_features_stepwise_vs_reference_golden(name):
    """irrelevant_features=True (Tuple spaces): pairs of actions in, pairs of states out; the
    reference's own trajectories incl. masked resets, and all four generators' end states."""
    from mdp_playground_amd import _capi as capi
    g = gu.load(name)
    E, T, _ = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_state"].astype(np.int64))
    assert np.array_equal(env.get_rng_streams(0), g["rng_env"])
    # (state and increment; the stale 32-bit buffer word of a generator that never draws 32-bit
    # integers is not carried to the device)
    assert np.array_equal(env.get_rng_streams(capi.STREAM_SPACE)[:, :4], g["rng_space"][:, :4])
    assert np.array_equal(env.get_rng_streams(capi.STREAM_SPACE_IRR)[:, :4], g["rng_space_irr"][:, :4])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy(), g["obs"][:, t].astype(np.int64)), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra].astype(np.int64))
    st = env.get_augmented_state()
    assert np.array_equal(st["curr_state"], env._obs.cpu().numpy())
    env.close()


@pytest.mark.parametrize("name", [n for n in gu.IRRELEVANT if gu.CASES[n]["reset"] == "on_done"])
def test_irrelevant_features_fused_rollout_vs_reference_golden(name):
    g = gu.load(name)
    env = _venv(autoreset="same_step", **_seeds_or_cfg(name))
    acts = torch.as_tensor(np.ascontiguousarray(g["action"].transpose(1, 0, 2).astype(np.int32)), device=env.device)
    obs, rew, term, trunc = env.rollout(acts)
    exp = g["obs"].astype(np.int64).copy()
    ra = g["reset_after"]
    exp[ra] = g["reset_obs"].astype(np.int64)[ra]
    assert np.array_equal(obs.cpu().numpy().transpose(1, 0, 2), exp)
    assert np.array_equal(term.cpu().numpy().T, g["done"])
    assert np.array_equal(rew.cpu().numpy().T, g["reward"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_irrelevant_features_2048_envs_vs_oracle(rng):
    """One shared MDP with an irrelevant sub-space and both noises, 2 048 instances with their own
    streams (numpy PCG64, or Philox keyed by the global env id), fused rollout with same-step
    autoreset and final observations through single steps; every 5th instance against the oracle."""
    from mdp_playground_amd import _capi as capi
    cfg = dict(gu.CASES["d_irr_noise"]["config"], seed=21)
    N, T, off = 2048, 80, 4096
    kw = dict(rng="philox", philox_seed=5, env_id_offset=off) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    r = np.random.default_rng(8)
    acts = np.stack([r.integers(0, 8, size=(T, N)), r.integers(0, 5, size=(T, N))], axis=2).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    for i in range(0, N, 5):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(5, off + i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            o.set_rng_irr(env.seeded_streams[capi.STREAM_SPACE_IRR][i])
        assert list(o.reset()) == [int(x) for x in init[i]]
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i], exp), i
        assert np.array_equal(term[:, i], ed), i
        assert np.array_equal(rew[:, i], er.astype(np.float32)), i
        if rng == "numpy":
            assert np.array_equal(o.get_rng_irr()[:4], env.get_rng_streams(capi.STREAM_SPACE_IRR)[i][:4])
    env.close()


# ----------------------------------------------------------------------------- grid envs
@pytest.mark.parametrize("name", gu.GRID)
def test_grid_stepwise_vs_reference_golden(name):
    """Grid envs against the reference's own trajectories: clipped moves, out-of-space actions as
    noops (status bit), noisy actions re-drawn from the action space's generator, dense / sparse
    rewards, the latched target flag, masked resets from the feature space's generator."""
    from mdp_playground_amd import _capi as capi
    g = gu.load(name)
    E, T, G = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_state"])
    assert np.array_equal(env.get_rng_streams(0), g["rng_env"])
    bad_seen = False
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy(), g["obs"][:, t]), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        bad = (np.abs(g["action"][:, t]).sum(axis=1) > 1) | (np.abs(g["action"][:, t]).max(axis=1) > 1)
        assert np.array_equal((env.status() & capi.STATUS_BAD_ACTION) != 0, bad), (name, t)
        bad_seen |= bool(bad.any())
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra])
    assert bad_seen == ("bad_action_every" in gu.CASES[name])
    assert np.array_equal(env.get_rng_streams(capi.STREAM_ACTION)[:, :4] == g["rng_action"][:, :4],
                          np.ones((E, 4), bool)) == (not gu.CASES[name]["config"].get("transition_noise"))
    st = env.get_augmented_state()
    assert np.array_equal(st["curr_state"], env._obs.cpu().numpy())
    env.set_augmented_state(st)
    env.close()


@pytest.mark.parametrize("name", [n for n in gu.GRID if gu.CASES[n]["reset"] == "on_done"])
def test_grid_fused_rollout_vs_reference_golden(name):
    g = gu.load(name)
    env = _venv(autoreset="same_step", **_seeds_or_cfg(name))
    acts = torch.as_tensor(np.ascontiguousarray(g["action"].transpose(1, 0, 2).astype(np.int32)), device=env.device)
    obs, rew, term, trunc = env.rollout(acts)
    exp = g["obs"].copy()
    ra = g["reset_after"]
    exp[ra] = g["reset_obs"][ra]
    assert np.array_equal(obs.cpu().numpy().transpose(1, 0, 2), exp)
    assert np.array_equal(term.cpu().numpy().T, g["done"])
    assert np.array_equal(rew.cpu().numpy().T, g["reward"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_grid_4096_envs_vs_oracle(rng):
    """One grid config with both noises and an irrelevant grid, 4 096 instances with their own
    streams (numpy PCG64, or Philox keyed by the global env id), fused rollout with same-step
    autoreset; every 7th instance against the oracle, incl. the three generators' end states."""
    from mdp_playground_amd import _capi as capi
    cfg = dict(state_space_type="grid", grid_shape=(5, 6), reward_function="move_to_a_point", make_denser=True,
               target_point=[2, 2], irrelevant_features=True, transition_noise=0.25, reward_noise=0.1,
               reward_scale=2.0, term_state_reward=0.5, seed=13)
    N, T, off = 4096, 120, 8192
    kw = dict(rng="philox", philox_seed=3, env_id_offset=off) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    r = np.random.default_rng(2)
    acts = np.zeros((T, N, 4), np.int32)
    which = r.integers(0, 4, size=(T, N))
    val = r.integers(-1, 2, size=(T, N))
    np.put_along_axis(acts, which[..., None], val[..., None], axis=2)
    acts[r.random((T, N)) < 0.02] = 1                      # outside the action space: noop
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    assert term.any()
    ends = [env.get_rng_streams(s) for s in (capi.STREAM_ENV, capi.STREAM_SPACE, capi.STREAM_ACTION)] if rng == "numpy" else None
    for i in range(0, N, 7):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(3, off + i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i], env.seeded_streams[capi.STREAM_ACTION][i])
        assert np.array_equal(o.reset(), init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i], exp), i
        assert np.array_equal(term[:, i], ed), i
        assert np.array_equal(rew[:, i], er.astype(np.float32)), i
        if rng == "numpy":
            w = o.get_rng()
            assert all(np.array_equal(w[k][:4], ends[k][i][:4]) for k in range(3)), i
            assert np.array_equal(w[2][4:], ends[2][i][4:]), i          # numpy's buffered 32-bit half
    env.close()


GRID_FAST = {
    "dense_2d": (dict(grid_shape=(8, 8), make_denser=True, target_point=[5, 5], reward_scale=3.0,
                      term_state_reward=-0.25), {}, 4100),
    "sparse_4d_i32_every2": (dict(grid_shape=(5, 7), make_denser=False, target_point=[1, 2], irrelevant_features=True,
                                  reward_every_n_steps=2, reward_shift=0.5, dtype_s=np.int32), {}, 1000),
    "dense_2d_horizon": (dict(grid_shape=(6, 6), make_denser=True, target_point=[0, 0]), dict(max_episode_steps=9), 1024),
    "dense_2d_noreset": (dict(grid_shape=(4, 9), make_denser=True, target_point=[3, 8], term_state_reward=1.0),
                         dict(autoreset="disabled"), 777),
    # noise: the env stream (noise trigger, reward normal) and the action space's stream (re-drawn action)
    "pn_dense_2d": (dict(grid_shape=(8, 8), make_denser=True, target_point=[5, 5], transition_noise=0.3,
                         term_state_reward=-0.25), {}, 1000),
    "rn_sparse_2d": (dict(grid_shape=(5, 7), make_denser=False, target_point=[2, 4], reward_noise=0.4,
                          reward_scale=2.0, reward_shift=0.5), {}, 512),
    "pn_rn_4d_every2": (dict(grid_shape=(6, 9), make_denser=True, target_point=[1, 7], irrelevant_features=True,
                             transition_noise=0.2, reward_noise=0.1, reward_every_n_steps=2), dict(max_episode_steps=11), 640),
}


@pytest.mark.parametrize("variant", sorted(GRID_FAST))
def test_grid_fast_rollout_kernel_vs_oracle(variant):
    """k_grid_rollout_fast (quiet numpy-stream grid handles): start cells drawn ahead into a register
    queue and un-drawn at the end of the launch.  Every env against its oracle over several launches
    of different lengths — observations, rewards, flags, truncation, the terminal observations, and
    the feature-space generator's state after every launch (the un-draw must leave it exactly where
    the reference's would be)."""
    from mdp_playground_amd import _capi as capi
    cfg_extra, env_kw, N = GRID_FAST[variant]
    cfg = dict(state_space_type="grid", reward_function="move_to_a_point", seed=31, **cfg_extra)
    kw = dict(autoreset="same_step")
    kw.update(env_kw)
    env = _venv(num_envs=N, **kw, **cfg)
    assert env.rollout_kernel_name(8).startswith("k_grid_rollout_fast<")
    G = len(env.mdps[0].grid_shape)
    horizon = env_kw.get("max_episode_steps", 0)
    auto = kw["autoreset"] == "same_step"
    r = np.random.default_rng(12)
    oracles = []
    init = env._obs.cpu().numpy().copy()
    for i in range(0, N, 9):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i], env.seeded_streams[capi.STREAM_ACTION][i])
        assert np.array_equal(o.reset(), init[i])
        oracles.append((i, o, [0]))
    for K in (1, 7, 64, 33):
        acts = np.zeros((K, N, G), np.int32)
        toward = r.random((K, N)) < 0.5
        np.put_along_axis(acts, r.integers(0, G, size=(K, N, 1)), r.integers(-1, 2, size=(K, N, 1)).astype(np.int32), axis=2)
        acts[r.random((K, N)) < 0.01] = 1                       # outside the action space
        del toward
        at = torch.as_tensor(acts, device=env.device)
        if K == 1:
            o1, r1, t1, tr1, info = env.step(at[0])
            obs, rew, term, trunc = o1[None].cpu().numpy(), r1[None].cpu().numpy(), t1[None].cpu().numpy(), tr1[None].cpu().numpy()
            fin = info["final_obs"].cpu().numpy() if auto else None
        else:
            obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(at))
            fin = None
        sp_end = env.get_rng_streams(capi.STREAM_SPACE)
        env_end, act_end = env.get_rng_streams(capi.STREAM_ENV), env.get_rng_streams(capi.STREAM_ACTION)
        for i, o, ep in oracles:
            for t in range(K):
                st, rr, d = o.step(acts[t, i])
                ep[0] += 1
                tr = bool(horizon) and ep[0] >= horizon
                assert d == bool(term[t, i]) and tr == bool(trunc[t, i]), (variant, K, i, t)
                assert np.float32(rr) == rew[t, i], (variant, K, i, t)
                if auto and (d or tr):
                    if fin is not None:
                        assert np.array_equal(fin[i], st), (variant, K, i)
                    st = o.reset(explicit=False)
                    ep[0] = 0
                assert np.array_equal(obs[t, i], st), (variant, K, i, t)
            assert np.array_equal(o.get_rng()[1][:4], sp_end[i][:4]), (variant, K, i)
            assert np.array_equal(o.get_rng()[0][:4], env_end[i][:4]), (variant, K, i)
            assert np.array_equal(o.get_rng()[2], act_end[i]), (variant, K, i)       # incl. the buffered 32-bit half
    env.close()


# ----------------------------------------------------------------------------- continuous
# move_along_a_line: the reference takes the line's direction from LAPACK's float32 SVD, so its reward
# is only defined up to that library's rounding (its own tests compare with atol=1e-5, see
# tests/test_mdp_playground.py:60-63); the device computes the direction in float64.  States,
# terminal flags and everything else stay bit-exact.
LINE_ATOL = 1e-5


def _rewards_match(name, got, exp):
    cfg = gu.CASES[name]["config"]
    if cfg.get("reward_function") == "move_along_a_line":
        return np.allclose(got, exp, rtol=0, atol=LINE_ATOL * cfg.get("reward_scale", 1.0))
    return np.array_equal(got, exp)


@pytest.mark.parametrize("name", gu.CONTINUOUS)
def test_continuous_stepwise_vs_reference_golden(name):
    g = gu.load(name)
    E, T, D = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_state"])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t], device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy().view(np.uint32), g["obs"][:, t].view(np.uint32)), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert _rewards_match(name, rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra])
    sd = env.get_augmented_state()["state_derivatives"]
    assert np.array_equal(sd.view(np.uint32), g["sd"][:, -1].view(np.uint32)) or g["reset_after"][:, -1].any()
    env.close()


@pytest.mark.parametrize("name", [n for n in gu.CONTINUOUS if gu.CASES[n]["reset"] == "on_done"])
def test_continuous_fused_rollout_vs_reference_golden(name):
    g = gu.load(name)
    E, T, D = g["action"].shape
    env = _venv(autoreset="same_step", **_seeds_or_cfg(name))
    acts = torch.as_tensor(np.ascontiguousarray(g["action"].transpose(1, 0, 2)), device=env.device)
    obs, rew, term, trunc = env.rollout(acts)
    exp = g["obs"].copy()
    ra = g["reset_after"]
    exp[ra] = g["reset_obs"][ra]
    assert np.array_equal(obs.cpu().numpy().transpose(1, 0, 2).view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(term.cpu().numpy().T, g["done"])
    assert _rewards_match(name, rew.cpu().numpy().T, g["reward"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
@pytest.mark.parametrize("dense", [True, False])
def test_default_target_point_2048_envs_vs_oracle(rng, dense):
    """No target_point in the config: the reference's float64 zeros over every dimension
    (rl_toy_env.py:652-654) -- float64 distances, target latch and (dense) float64 reward through the delay line,
    noise and the affine map; a sparse reward turns np.float32 at the action penalty.  Fused rollout with same-step
    autoreset, then single steps, every 7th env against the oracle: states bit-exact, rewards equal as float32,
    a get/set_augmented_state round trip of the float64 delay line."""
    cfg = dict(gu.CASES["c_default_target" if dense else "c_default_target_sparse"]["config"], seed=12)
    cfg.update(delay=3, target_radius=1.2)
    N, T, T1 = 2048, 50, 5
    kw = dict(rng="philox", philox_seed=23) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=17, **kw, **cfg)
    assert env.mdps[0].target_default and "rollout" not in env.rollout_kernel_name(T)      # the general kernel
    D = cfg["state_space_dim"]
    acts = np.random.default_rng(4).uniform(-1, 1, size=(T + T1, N, D)).astype(np.float32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts[:T], device=env.device)))
    st = env.get_augmented_state()
    assert st["reward_buffer"].shape == (N, 3) and not st["reward_buffer_is32"].any() if dense else True
    twin = _venv(num_envs=N, autoreset="same_step", max_episode_steps=17, **kw, **cfg)
    twin._lib.mdpp_tick(twin._h, T, None)                 # the step counter first: Philox keys, and the head of the delay line
    twin.set_augmented_state(st)
    if rng == "numpy":
        for s_ in (0, 1):
            twin._put_stream(s_, env.get_rng_streams(s_))
    tail = []
    for t in range(T, T + T1):
        a = torch.as_tensor(acts[t], device=env.device)
        o1, r1, d1, tr1, _ = env.step(a)
        o2, r2, d2, tr2, _ = twin.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), t
        tail.append((o1.cpu().numpy().copy(), r1.cpu().numpy().copy(), d1.cpu().numpy().copy()))
    assert term.any() and trunc.any()
    for i in range(0, N, 7):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(23, i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(T + T1):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            etr = n >= 17
            if ed or etr:
                eo = o.reset(explicit=False)
                n = 0
            go, gr, gd = (obs[t, i], rew[t, i], term[t, i]) if t < T else (tail[t - T][0][i], tail[t - T][1][i], tail[t - T][2][i])
            assert np.array_equal(np.asarray(eo).view(np.uint32), go.view(np.uint32)), (i, t)
            assert np.float32(er) == gr and bool(ed) == bool(gd), (i, t, er, gr)
    env.close(); twin.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
@pytest.mark.parametrize("shape", ["d4_o1_l10", "d2_o2_l16", "d4_o2_l3"])
def test_line_rollout_kernel_vs_general_kernel_and_oracle(shape, rng):
    """k_continuous_line_rollout (round 3: the line reward's common shape -- every dimension relevant, no noise, delay 0 --
    with the window in LDS and its moments carried) against k_continuous_step (NO_CFAST) on EVERY env of 16 384: states,
    flags, truncations and post-reset states bit-equal, rewards within 1e-6 (running moments against recomputed ones: a few
    float64 ulps before the float32 rounding); launches of different lengths, state carried across them and handed between
    the two kernels through the HBM history; a sample of envs against the oracle within the upstream tolerance."""
    D, order, L = {"d4_o1_l10": (4, 1, 10), "d2_o2_l16": (2, 2, 16), "d4_o2_l3": (4, 2, 3)}[shape]
    cfg = dict(state_space_type="continuous", state_space_dim=D, transition_dynamics_order=order, inertia=1, time_unit=0.5 if order == 2 else 1,
               delay=0, sequence_length=L, reward_scale=1.5, reward_shift=0.25, action_space_max=1, state_space_max=6,
               reward_function="move_along_a_line", seed=8)
    N = 16384
    kw = dict(rng="philox", philox_seed=29) if rng == "philox" else {}
    a = _venv(num_envs=N, autoreset="same_step", max_episode_steps=37, **kw, **cfg)
    b = _venv(num_envs=N, autoreset="same_step", max_episode_steps=37, **kw, **cfg)
    b.set_kernel_options("NO_CFAST")
    assert a.rollout_kernel_name(64).startswith("k_continuous_line_rollout<") and b.rollout_kernel_name(64).startswith("k_continuous_step<")
    g = torch.Generator(device=a.device)
    g.manual_seed(3)
    init = a._obs.cpu().numpy().copy()
    all_acts, all_out = [], []
    for j, K in enumerate((64, 7, 33, 50)):
        acts = torch.rand((K, N, D), generator=g, device=a.device) * 2.4 - 1.2       # (some actions outside the box: "stay")
        oa, ra, ta, tra = a.rollout(acts)
        if j == 2:                        # hand the envs over: b runs this launch on the line kernel, a on the general one
            a.set_kernel_options("NO_CFAST"); b.set_kernel_options()
        ob, rb, tb, trb = b.rollout(acts)
        if j == 2:
            a.set_kernel_options(); b.set_kernel_options("NO_CFAST")
        assert torch.equal(oa, ob) and torch.equal(ta, tb) and torch.equal(tra, trb), (shape, j)
        assert float((ra - rb).abs().max()) <= 1e-6, (shape, j, float((ra - rb).abs().max()))
        all_acts.append(acts.cpu().numpy()); all_out.append((oa.cpu().numpy(), ra.cpu().numpy(), tra.cpu().numpy()))
    assert any(o[2].any() for o in all_out)
    acts = np.concatenate(all_acts); obs = np.concatenate([o[0] for o in all_out]); rew = np.concatenate([o[1] for o in all_out])
    trn = np.concatenate([o[2] for o in all_out])
    for i in range(5, N, 1637):
        o = _oracle_for(a, i)
        if rng == "philox":
            o.set_philox(29, i)
        else:
            o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(acts.shape[0]):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            if n >= 37:
                assert trn[t, i]
                eo = o.reset(explicit=False); n = 0
            assert np.array_equal(np.asarray(eo).view(np.uint32), obs[t, i].view(np.uint32)), (shape, i, t)
            assert abs(float(rew[t, i]) - er) <= LINE_ATOL * 1.5, (shape, i, t, rew[t, i], er)
    a.close(); b.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_line_reward_with_five_to_eight_relevant_dimensions_vs_oracle(rng):
    """move_along_a_line with more than 4 relevant dimensions (VERDICT r2 "missing"; reference golden `c_line_6of8` in the
    stepwise test): k_continuous_step<12, OMAX, PHILOX, NL = 8> -- rows of 8 in the line history, an 8 x 8 float64 scatter
    matrix -- on 512 envs: 7 relevant of 10 dimensions, order 2, delay 2, reward noise, truncation at 29 steps, fused
    rollouts of 50 and 33 steps then single steps; every 7th env against the oracle (states and flags bit-exact, rewards
    within the upstream tolerance); 8 relevant of 8; and more than 12 state dimensions are refused at construction."""
    from mdp_playground_amd import _capi as capi
    cfg = dict(state_space_type="continuous", state_space_dim=10, irrelevant_features=True,
               relevant_indices=[0, 1, 3, 4, 6, 8, 9], transition_dynamics_order=2, inertia=1.0, time_unit=0.5,
               state_space_max=8, action_space_max=1, delay=2, sequence_length=6, reward_noise=0.05, reward_scale=2.0,
               reward_shift=0.5, reward_function="move_along_a_line", seed=17)
    N, T = 512, 50 + 33 + 6
    kw = dict(rng="philox", philox_seed=5) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=29, **kw, **cfg)
    assert env.rollout_kernel_name(50).endswith("NL=8>")
    rs = np.random.default_rng(4)
    acts = rs.uniform(-1.1, 1.1, size=(T, N, 10)).astype(np.float32)
    acts[20:40] = acts[20]                                     # a straight stretch: rewards near zero, tiny singular-value gap
    init = env._obs.cpu().numpy().copy()
    outs = [env.rollout(torch.as_tensor(acts[:50], device=env.device)), env.rollout(torch.as_tensor(acts[50:83], device=env.device))]
    obs = np.concatenate([o[0].cpu().numpy() for o in outs]); rew = np.concatenate([o[1].cpu().numpy() for o in outs])
    trn = np.concatenate([o[3].cpu().numpy() for o in outs])
    for t in range(83, T):
        o1, r1, te, tr, _ = env.step(torch.as_tensor(acts[t], device=env.device))
        obs = np.concatenate([obs, o1.cpu().numpy()[None]]); rew = np.concatenate([rew, r1.cpu().numpy()[None]])
        trn = np.concatenate([trn, tr.cpu().numpy()[None]])
    assert trn.any()
    for i in range(0, N, 7):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(5, i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(T):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            if n >= 29:
                assert trn[t, i]
                eo = o.reset(explicit=False); n = 0
            assert np.array_equal(np.asarray(eo).view(np.uint32), obs[t, i].view(np.uint32)), (i, t)
            assert abs(float(rew[t, i]) - er) <= LINE_ATOL * 2.0 * 1.5, (i, t, rew[t, i], er)
    env.close()
    cfg8 = dict(state_space_type="continuous", state_space_dim=8, transition_dynamics_order=1, inertia=1.0, time_unit=1.0,
                state_space_max=5, action_space_max=1, delay=0, sequence_length=20, reward_function="move_along_a_line", seed=2)
    env = _venv(num_envs=256, autoreset="same_step", max_episode_steps=40, **kw, **cfg8)
    a8 = rs.uniform(-1, 1, size=(64, 256, 8)).astype(np.float32)
    init = env._obs.cpu().numpy().copy()
    ob, rw, _, tr8 = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(a8, device=env.device)))
    for i in range(0, 256, 31):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(5, i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(64):
            eo, er, _, ed = o.step(a8[t, i])
            n += 1
            if n >= 40:
                eo = o.reset(explicit=False); n = 0
            assert np.array_equal(np.asarray(eo).view(np.uint32), ob[t, i].view(np.uint32)), (i, t)
            assert abs(float(rw[t, i]) - er) <= LINE_ATOL * 1.5, (i, t, rw[t, i], er)
    env.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
@pytest.mark.parametrize("shape", ["n12_of_14", "n9_of_9", "n7_of_16", "n32_of_32"])
def test_line_reward_beyond_eight_relevant_dimensions_vs_oracle(shape, rng):
    """move_along_a_line has no dimension limit in the reference (rl_toy_env.py:1865-1910, :2546-2576); rounds 2-5 refused more
    than 8 relevant dimensions (VERDICT r4 / r5 "missing").  c_line_reward_big: the same fit with its n x n matrices in an HBM
    workspace -- 12 relevant of 14 dimensions (order 2, delay, reward noise, truncation), 9 of 9, 7 of 16 (rows of 8 no longer
    need state_space_dim <= 12) and the maximum, 32 of 32 with a window of 64 states: fused rollouts and single steps, sampled
    envs against the oracle (states and flags bit-exact, rewards within the upstream tolerance; reference golden
    `c_line_12of14` in the stepwise test)."""
    D, rel, order, L, delay = {"n12_of_14": (14, [0, 1, 2, 4, 5, 6, 7, 8, 10, 11, 12, 13], 2, 6, 2), "n9_of_9": (9, list(range(9)), 1, 12, 0),
                               "n7_of_16": (16, [0, 3, 5, 8, 9, 12, 15], 1, 5, 1), "n32_of_32": (32, list(range(32)), 1, 64, 0)}[shape]
    cfg = dict(state_space_type="continuous", state_space_dim=D, transition_dynamics_order=order, inertia=1.0,
               time_unit=0.5 if order == 2 else 1.0, state_space_max=8, action_space_max=1, delay=delay, sequence_length=L,
               reward_scale=2.0, reward_shift=0.5, reward_function="move_along_a_line", seed=17)
    if len(rel) != D:
        cfg.update(irrelevant_features=True, relevant_indices=rel)
    if shape == "n12_of_14":
        cfg.update(reward_noise=0.05)
    N = 128 if D == 32 else 256
    T1, T2, T3 = (L + 20, 9, 4) if D == 32 else (50, 33, 6)
    T, tl = T1 + T2 + T3, 2 * L + 5
    kw = dict(rng="philox", philox_seed=5) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=tl, **kw, **cfg)
    assert env.rollout_kernel_name(T1).startswith("k_continuous_step<DMAX=%d," % (12 if D <= 12 else 16 if D <= 16 else 32))
    rs = np.random.default_rng(4)
    acts = rs.uniform(-1.1, 1.1, size=(T, N, D)).astype(np.float32)
    acts[20:20 + min(20, L + 4)] = acts[20]                    # a straight stretch: rewards near zero, tiny singular-value gap
    init = env._obs.cpu().numpy().copy()
    outs = [env.rollout(torch.as_tensor(acts[:T1], device=env.device)), env.rollout(torch.as_tensor(acts[T1:T1 + T2], device=env.device))]
    obs = np.concatenate([o[0].cpu().numpy() for o in outs]); rew = np.concatenate([o[1].cpu().numpy() for o in outs])
    trn = np.concatenate([o[3].cpu().numpy() for o in outs])
    for t in range(T1 + T2, T):
        o1, r1, te, tr, _ = env.step(torch.as_tensor(acts[t], device=env.device))
        obs = np.concatenate([obs, o1.cpu().numpy()[None]]); rew = np.concatenate([rew, r1.cpu().numpy()[None]])
        trn = np.concatenate([trn, tr.cpu().numpy()[None]])
    assert not (env.status() & 0x80000000).any() and (trn.any() or T < tl)     # (actions beyond the box are 'stay' steps: BAD_ACTION is expected)
    assert np.abs(rew).max() > 1e-3
    for i in range(0, N, 37):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(5, i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(T):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            if n >= tl:
                assert trn[t, i]
                eo = o.reset(explicit=False); n = 0
            assert np.array_equal(np.asarray(eo).view(np.uint32), obs[t, i].view(np.uint32)), (shape, i, t)
            assert abs(float(rew[t, i]) - er) <= LINE_ATOL * 2.0 * 1.5 * max(1.0, len(rel) / 8.0), (shape, i, t, rew[t, i], er)
    env.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_continuous_line_reward_1024_envs_vs_oracle(rng):
    """reward_function move_along_a_line on 1024 envs (5 relevant-of-6 dims would not fit: 3 of 6
    here), random walks with straight stretches, delay, reward noise, terminal hypercubes, same-step
    autoreset, fused rollout then single steps; every 9th env against its own oracle instance (which
    gets mean and singular vector from numpy like the reference).  Everything but the reward is
    bit-exact; the reward within LINE_ATOL, and within 1e-6 wherever the two largest singular
    values are not close (the direction is then well-conditioned)."""
    from oracle import oracle as ora
    cfg = dict(gu.CASES["c_line_irr"]["config"], seed=21)
    N, T, T1 = 1024, 60, 6
    kw = dict(rng="philox", philox_seed=17) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    D = cfg["state_space_dim"]
    g = np.random.default_rng(8)
    acts = g.uniform(-1, 1, size=(T + T1, N, D)).astype(np.float32)
    for t in range(1, T + T1):                      # straight stretches: repeat the previous action
        keep = (t // 7) % 2 == 1
        if keep:
            acts[t] = acts[t - 1]
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts[:T], device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    for t in range(T, T + T1):
        o1, r1, d1, _, _ = env.step(torch.as_tensor(acts[t], device=env.device))
        obs = np.concatenate([obs, o1.cpu().numpy()[None]]); rew = np.concatenate([rew, r1.cpu().numpy()[None]])
        term = np.concatenate([term, d1.cpu().numpy()[None]])
    worst = 0.0
    for i in range(0, N, 9):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(17, i)
        else:
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i].view(np.uint32), exp.view(np.uint32)), i
        assert np.array_equal(term[:, i], ed), i
        diff = np.abs(rew[:, i].astype(np.float64) - er)
        worst = max(worst, float(diff.max()))
        assert diff.max() <= LINE_ATOL * cfg["reward_scale"], (i, diff.max())
        assert np.quantile(diff, 0.99) <= 1e-6, (i, np.quantile(diff, 0.99))
    print("move_along_a_line: worst |reward - oracle| =", worst)
    env.close()


# ----------------------------------------------------------------------------- fast rollout kernel
FAST_VARIANTS = {
    "cfg2_k100": (dict(gu.CASES["d_cfg2"]["config"], seed=11), {}, 100),
    "cfg1_k7": (dict(gu.CASES["d_cfg1"]["config"], seed=5), {}, 7),
    "s6_l2_nonpow2": (dict(state_space_type="discrete", action_space_type="discrete",
                           state_space_size=6, action_space_size=6, delay=0, sequence_length=2,
                           reward_scale=2.5, reward_shift=-0.75, term_state_reward=-1.5, seed=2), {}, 64),
    "s16_l1_trunc": (dict(state_space_type="discrete", action_space_type="discrete",
                          state_space_size=16, action_space_size=16, delay=1, sequence_length=1,
                          terminal_state_density=0.0625, seed=4), dict(max_episode_steps=5), 61),
    "diam2_l3": (dict(state_space_type="discrete", action_space_type="discrete",
                      state_space_size=12, action_space_size=6, diameter=2, delay=2,
                      sequence_length=3, reward_every_n_steps=1, seed=9), {}, 80),
    "cfg2_int32": (dict(gu.CASES["d_cfg2"]["config"], seed=1, dtype_o=np.int32), {}, 40),
}


def _oracle_autoreset_rollout(o, acts, max_steps):
    """Reference loop: step(); if done (or TimeLimit-truncated): reset()."""
    T = len(acts)
    obs = np.zeros(T, np.int64); rew = np.zeros(T); term = np.zeros(T, bool); trunc = np.zeros(T, bool)
    n = 0
    for t in range(T):
        ob, r, d = o.step(int(acts[t]))
        n += 1
        tr = bool(max_steps) and n >= max_steps
        if d or tr:
            ob = o.reset()
            n = 0
        obs[t], rew[t], term[t], trunc[t] = ob, r, d, tr
    return obs, rew, term, trunc


@pytest.mark.parametrize("variant", sorted(FAST_VARIANTS))
@pytest.mark.parametrize("fused,N", [(True, 1000), (True, 1024), (False, 1000), ("rollout_k1", 1000), ("mixed", 1000)])
def test_discrete_fast_kernel_vs_oracle(variant, fused, N):
    """N = 1000: partial last block, RNG drawn by the env lanes; N = 1024 and >= 32 fused steps:
    the helper-wave variant (start states produced by partner waves through an LDS ring).  Single steps (fused False):
    k_discrete_step1, the kernel of a one-step launch (round 5); "rollout_k1": the rollout kernel with K = 1, which served
    mdpp_step before; "mixed": single steps, a fused piece, single steps again -- the queue of start states is shared."""
    cfg, kw, T = FAST_VARIANTS[variant]
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    if fused == "rollout_k1":
        env.set_kernel_options("NO_STEP1")
        assert env.rollout_kernel_name(1).startswith("k_discrete_rollout_fast")
        fused = False
    else:
        assert env.rollout_kernel_name(1).startswith("k_discrete_step1<"), env.rollout_kernel_name(1)
    A = env.mdps[0].A
    acts = np.random.default_rng(3).integers(0, A, size=(T, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    if fused is True:
        obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
        obs, rew, term, trunc = (x.cpu().numpy() for x in (obs, rew, term, trunc))
    else:
        obs = np.zeros((T, N), np.int64); rew = np.zeros((T, N), np.float32)
        term = np.zeros((T, N), bool); trunc = np.zeros((T, N), bool)
        k0, k1 = (T // 3, T - T // 3) if fused == "mixed" else (T, T)       # [k0, k1): one fused launch
        for t in list(range(k0)) + list(range(k1, T)):
            if t == k1 and k1 > k0:
                fo = env.rollout(torch.as_tensor(acts[k0:k1], device=env.device))
                obs[k0:k1], rew[k0:k1], term[k0:k1], trunc[k0:k1] = (x.cpu().numpy() for x in fo)
            o, r, te, tr, info = env.step(torch.as_tensor(acts[t], device=env.device))
            obs[t], rew[t], term[t], trunc[t] = (x.cpu().numpy() for x in (o, r, te, tr))
    end_env = env.get_rng_streams(0)
    for i in range(0, N, 23):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert o.reset() == int(init[i])
        eo, er, et, etr = _oracle_autoreset_rollout(o, acts[:, i], kw.get("max_episode_steps", 0))
        assert np.array_equal(obs[:, i], eo), (variant, i)
        assert np.array_equal(term[:, i], et) and np.array_equal(trunc[:, i], etr), (variant, i)
        assert np.array_equal(rew[:, i], er.astype(np.float32)), (variant, i)
        # the draw-ahead queue must be invisible: reported stream state == reference's
        assert np.array_equal(o.get_rng()[0][:4], end_env[i][:4]), (variant, i)
    env.close()


def test_discrete_fast_kernel_masked_reset_and_reseed():
    """reset(mask) consumes queued draws in stream order; reset(seed=) voids the queue."""
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=21)
    N, T = 512, 40
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    rng = np.random.default_rng(8)
    acts = rng.integers(0, 8, size=(T, N)).astype(np.int32)
    oracles = {}
    for i in range(0, N, 31):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        o.reset()
        oracles[i] = o
    env.rollout(torch.as_tensor(acts, device=env.device))
    for i, o in oracles.items():
        _oracle_autoreset_rollout(o, acts[:, i], 0)
    mask = rng.random(N) < 0.5
    ob, _ = env.reset(mask=torch.as_tensor(mask, device=env.device))
    ob = ob.cpu().numpy()
    for i, o in oracles.items():
        if mask[i]:
            assert o.reset() == ob[i]
    obs2, *_ = env.rollout(torch.as_tensor(acts, device=env.device))
    obs2 = obs2.cpu().numpy()
    for i, o in oracles.items():
        eo, *_ = _oracle_autoreset_rollout(o, acts[:, i], 0)
        assert np.array_equal(obs2[:, i], eo), i
    # re-seed: env i restarts from PCG64(SeedSequence(777 + i))
    ob, _ = env.reset(seed=777)
    ob = ob.cpu().numpy()
    obs3, *_ = env.rollout(torch.as_tensor(acts, device=env.device))
    obs3 = obs3.cpu().numpy()
    from mdp_playground_amd import mdp as mdp_mod
    for i in oracles:
        o = _oracle_for(env, i)
        o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator(777 + i)), env.seeded_streams[1][i])
        assert o.reset() == ob[i]
        eo, *_ = _oracle_autoreset_rollout(o, acts[:, i], 0)
        assert np.array_equal(obs3[:, i], eo), i
    env.close()


# ----------------------------------------------------------------------------- Philox streams
@pytest.mark.parametrize("fused", [True, False])
def test_discrete_philox_vs_oracle(fused):
    """rng='philox': stateless Philox4x32-10 keyed by (seed, GLOBAL env id, tick, stream).  Its
    Gaussians are float32 Box-Muller pairs built from IEEE-exact operations only (mdpp_rng.hpp
    philox_box_muller == oracle/np_random.c np_philox_box_muller), so rewards are bit-exact too."""
    cfg = dict(gu.CASES["d_cfg2_noise"]["config"], seed=6)
    N, T, off = 700, 50, 4096
    env = _venv(num_envs=N, autoreset="same_step", rng="philox", env_id_offset=off, philox_seed=99, **cfg)
    acts = np.random.default_rng(2).integers(0, 8, size=(T, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    if fused:
        obs, rew, term, _ = env.rollout(torch.as_tensor(acts, device=env.device))
        obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    else:
        obs = np.zeros((T, N), np.int64); rew = np.zeros((T, N), np.float32); term = np.zeros((T, N), bool)
        for t in range(T):
            o, r, te, _, _ = env.step(torch.as_tensor(acts[t], device=env.device))
            obs[t], rew[t], term[t] = o.cpu().numpy(), r.cpu().numpy(), te.cpu().numpy()
    for i in range(0, N, 13):
        o = _oracle_for(env, i)
        o.set_philox(99, off + i)
        assert o.reset() == int(init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(obs[:, i], eo), i
        assert np.array_equal(term[:, i], ed), i
        assert np.array_equal(rew[:, i], er.astype(np.float32)), i
    env.close()


@pytest.mark.parametrize("shape", ["s8", "s37_l2", "s255", "irr", "p_tiny", "p_big"])
def test_discrete_philox_noise_words_unaligned_ticks_vs_general_kernel_and_oracle(shape):
    """Philox streams, discrete noise: one word (one normal) per tick out of a block that serves four ticks
    (mdpp_rng.hpp philox_pnoise_*, PhiloxTickNormals; oracle/np_random.c np_philox_tick_*).  Launches that start and end
    inside a block (3 single steps, then rollouts of 50, 37 and 64 steps), small and large state spaces (the re-drawn
    state's index is a 32 x 64-bit product: exact floor(w (S - 1) / T) up to S = 255), noise probabilities from 1e-9 (below
    the threshold's resolution: never noisy) to 0.97, an irrelevant sub-space with its own noise words: the fused kernel
    (producer waves make the blocks) == the general kernel on every env, and every 11th env == the oracle."""
    base = dict(state_space_type="discrete", action_space_type="discrete", delay=1, sequence_length=2, reward_density=0.25,
                terminal_state_density=0.25, transition_noise=0.1, reward_noise=0.3, seed=5)
    cfg = dict(base, **{
        "s8": dict(state_space_size=8, action_space_size=8),
        "s37_l2": dict(state_space_size=37, action_space_size=5, reward_density=0.05, terminal_state_density=0.1),
        "s255": dict(state_space_size=255, action_space_size=3, sequence_length=1, reward_density=0.1, terminal_state_density=0.05),
        "irr": dict(state_space_size=[8, 6], action_space_size=[8, 6], irrelevant_features=True),
        "p_tiny": dict(state_space_size=8, action_space_size=8, transition_noise=1e-9),
        "p_big": dict(state_space_size=5, action_space_size=5, transition_noise=0.97),
    }[shape])
    N, off = 1024, 300
    kw = dict(rng="philox", philox_seed=1234, env_id_offset=off, autoreset="same_step", max_episode_steps=19)
    a = _venv(num_envs=N, **kw, **cfg)
    b = _venv(num_envs=N, **kw, **cfg)
    b.set_kernel_options("NO_PHILOX_FAST")
    assert a.rollout_kernel_name(50) != b.rollout_kernel_name(50)
    m = a.mdps[0]
    rs = np.random.default_rng(8)
    T = 3 + 50 + 37 + 64

    def acts_of(n):
        if shape == "irr":
            return np.stack([rs.integers(0, m.A, size=(n, N)), rs.integers(0, m.A_irr, size=(n, N))], axis=2).astype(np.int32)
        return rs.integers(0, m.A, size=(n, N)).astype(np.int32)
    acts = acts_of(T)
    init = a._obs.cpu().numpy().copy()
    outs = []
    for t in range(3):
        ra = a.step(torch.as_tensor(acts[t], device=a.device))
        rb = b.step(torch.as_tensor(acts[t], device=a.device))
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y), (shape, t)
        outs.append([x.cpu().numpy()[None] for x in ra[:4]])
    t0 = 3
    for n in (50, 37, 64):
        at = torch.as_tensor(acts[t0:t0 + n], device=a.device)
        ra, rb = a.rollout(at), b.rollout(at)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (shape, t0)
        outs.append([x.cpu().numpy() for x in ra])
        t0 += n
    obs, rew, term, trunc = (np.concatenate([o[j] for o in outs], axis=0) for j in range(4))
    assert (a.status() == 0).all() and (b.status() == 0).all()
    noisy_seen = 0
    for i in range(0, N, 11):
        o = _oracle_for(a, i)
        o.set_philox(1234, off + i)
        first = o.reset()
        assert np.array_equal(np.asarray(first), init[i])
        n = 0
        for t in range(T):
            if shape == "irr":
                eo, er, ed = o.step(acts[t, i])
            else:
                eo, er, ed = o.step(int(acts[t, i]))
            n += 1
            assert np.float32(er) == rew[t, i] and bool(ed) == bool(term[t, i]), (shape, i, t)
            tr = n >= 19
            assert tr == bool(trunc[t, i]), (shape, i, t)
            if ed or tr:
                eo = o.reset(explicit=False)
                n = 0
            assert np.array_equal(np.asarray(eo), obs[t, i]), (shape, i, t)
        noisy_seen += int(o.get_stats()[0][2] + o.get_stats()[1][2])
    if shape == "p_tiny":
        assert noisy_seen == 0
    elif shape != "irr":
        assert noisy_seen > 0
    a.close(); b.close()


def test_philox_box_muller_device_equals_oracle_bit_for_bit():
    """The Philox mode's Gaussian on 2^22 stream positions (incl. the first block of many envs and long
    streams of a few): device == oracle/np_random.c, every bit."""
    import ctypes as C
    from mdp_playground_amd import _capi as capi
    from oracle import oracle as ora
    lib = capi.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    for seed, env0, tick, stream, ne, npe in ((1234567890123, 0, 0, 0, 65536, 14), (99, 1 << 40, (1 << 33) + 5, 4, 64, 8192)):
        out = torch.empty((ne, npe), dtype=torch.float64, device=dev)
        rc = lib.mdpp_philox_normals(seed, env0, tick, stream, ne, npe, C.c_void_p(out.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0
        got = out.cpu().numpy()
        exp = ora.philox_normals(seed, env0, tick, stream, ne, npe)
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
        assert abs(got.mean()) < 0.01 and abs(got.std() - 1.0) < 0.01


def test_continuous_philox_vs_oracle_and_sharding_invariance():
    """Continuous cfg 5 with Philox streams: matches the oracle, and a job split into two shards
    (env_id_offset) reproduces the unsharded trajectories exactly (global-id keyed streams)."""
    cfg = dict(gu.CASES["c_cfg5"]["config"], seed=3)
    N, T = 512, 40
    rng = np.random.default_rng(4)
    acts = rng.uniform(-1, 1, size=(T, N, 12)).astype(np.float32)
    env = _venv(num_envs=N, autoreset="same_step", rng="philox", philox_seed=7, **cfg)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, _ = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    for i in range(0, N, 37):
        o = _oracle_for(env, i)
        o.set_philox(7, i)
        assert np.array_equal(o.reset(), init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(obs[:, i], eo), i
        assert np.array_equal(rew[:, i], er.astype(np.float32)), i
    env.close()
    half = N // 2
    parts = []
    for r in range(2):
        e = _venv(num_envs=half, autoreset="same_step", rng="philox", philox_seed=7,
                  env_id_offset=r * half, **cfg)
        a = torch.as_tensor(np.ascontiguousarray(acts[:, r * half:(r + 1) * half]), device=e.device)
        o, rw, _, _ = e.rollout(a)
        parts.append((o.cpu().numpy(), rw.cpu().numpy()))
        e.close()
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1), obs)
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), rew)


def test_numpy_streams_sharding_invariance_discrete():
    """rng='numpy': env i is seeded from its GLOBAL id, so 2 shards == 1 big batch."""
    cfg = dict(gu.CASES["d_cfg2_noise"]["config"], seed=12)
    N, T = 600, 64
    acts = np.random.default_rng(1).integers(0, 8, size=(T, N)).astype(np.int32)
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    obs, rew, term, _ = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    env.close()
    half = N // 2
    for r in range(2):
        e = _venv(num_envs=half, autoreset="same_step", env_id_offset=r * half, **cfg)
        a = torch.as_tensor(np.ascontiguousarray(acts[:, r * half:(r + 1) * half]), device=e.device)
        o, rw, _, _ = e.rollout(a)
        if r == 0:
            # env 0 keeps the reference's own post-construction space generator in both runs
            assert np.array_equal(o.cpu().numpy(), obs[:, :half])
            assert np.array_equal(rw.cpu().numpy(), rew[:, :half])
        else:
            assert np.array_equal(o.cpu().numpy(), obs[:, half:])
            assert np.array_equal(rw.cpu().numpy(), rew[:, half:])
        e.close()


# ----------------------------------------------------------------------------- image observations
@pytest.mark.parametrize("name", gu.IMAGE)
def test_image_observations_vs_reference_golden(name):
    """BASELINE cfg 4 shape (84x84 shift+rotate) and a 100x100 all-transforms case: every pixel of
    every observation equals what the reference (Pillow polygon + rotate) produced."""
    g = gu.load(name)
    E, T = g["action"].shape[:2]          # (i_irr: action pairs, one image per sub-space side by side)
    env = _venv(autoreset="same_step", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_obs"])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, info = env.step(a)
        obs, fin = obs.cpu().numpy(), info["final_obs"].cpu().numpy()
        d = g["done"][:, t]
        assert np.array_equal(term.cpu().numpy(), d)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32))
        assert np.array_equal(obs[~d], g["obs"][:, t][~d]), (name, t)
        assert np.array_equal(fin[d], g["obs"][:, t][d]), (name, t)          # terminal obs
        assert np.array_equal(obs[d], g["reset_obs"][:, t][d]), (name, t)    # first obs of next episode
    env.close()


IMG_CFGS = {
    # BASELINE cfg 4: fast renderer (k_image_obs_fast)
    "cfg4": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
                 action_space_size=8, delay=0, image_representations=True, image_width=84,
                 image_height=84, image_transforms="shift,rotate", image_sh_quant=1, image_ro_quant=1),
    # every transform, quantised shift/rotation, 100x100: general renderer (k_image_obs)
    "all100": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
                   action_space_size=8, delay=0, image_representations=True, image_width=100,
                   image_height=100, image_transforms="shift,scale,rotate,flip", image_sh_quant=2,
                   image_ro_quant=15, image_scale_range=(0.5, 1.5)),
    # 64x64, rotation and flips only (no shift): fast renderer, 4 stores per image
    "rot64": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=6,
                  action_space_size=6, delay=0, image_representations=True, image_width=64,
                  image_height=64, image_transforms="rotate,flip"),
    # an irrelevant sub-space: two images per observation from one stream (relevant, then irrelevant)
    "irr84": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=[8, 11],
                  action_space_size=[8, 11], irrelevant_features=True, delay=0, image_representations=True,
                  image_width=84, image_height=84, image_transforms="shift,rotate", image_sh_quant=1, image_ro_quant=1),
}


def _img_actions(cfg, shape, seed):
    r = np.random.default_rng(seed)
    A = cfg["action_space_size"]
    if isinstance(A, list):
        return np.stack([r.integers(0, a, size=shape) for a in A], axis=-1).astype(np.int32)
    return r.integers(0, A, size=shape).astype(np.int32)


@pytest.mark.parametrize("name", sorted(IMG_CFGS))
def test_image_fused_rollout_equals_single_steps(name):
    """mdpp_step_n on an image env runs batches of up to 64 steps (state kernel, serial draw kernel, per-image record
    kernel, persistent render kernel whose waves claim their images from counters), a long rollout starting with batches of 8,
    16 and 32; mdpp_step runs one step with the draw and the records fused.  Same arithmetic, different launch shapes:
    bit-identical images, rewards, flags and RNG end states.  K = 300: batches of 8, 16, 32, 64, 64, 64 and a ragged one of
    52 (cfg4; 40 elsewhere: one ragged batch)."""
    cfg = dict(IMG_CFGS[name], seed=5)
    N, K = (48, 300) if name == "cfg4" else (300, 40)
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    assert a.rollout_kernel_name(K) == "k_image_obs_wide" if name == "all100" else a.rollout_kernel_name(K).startswith("k_image_obs_fast<")
    acts = torch.as_tensor(_img_actions(cfg, (K, N), 2), device=a.device)
    obs, rew, term, trunc = a.rollout(acts)
    assert term.any() and not term.all()
    for t in range(K):
        o, r, te, tr, _ = b.step(acts[t])
        assert torch.equal(o, obs[t]), (name, t)
        assert torch.equal(r, rew[t]) and torch.equal(te, term[t]), (name, t)
    from mdp_playground_amd import _capi as capi
    assert np.array_equal(a.get_rng_streams(capi.STREAM_IMAGE), b.get_rng_streams(capi.STREAM_IMAGE))
    assert np.array_equal(a.get_rng_streams(capi.STREAM_ENV), b.get_rng_streams(capi.STREAM_ENV))
    a.close(); b.close()


@pytest.mark.parametrize("name", sorted(IMG_CFGS))
def test_image_batch_vs_oracle(name):
    """A few thousand images (all states, hundreds of distinct angles/shifts) against the oracle's
    draw + Pillow-exact rotate restatement, for the observation, the terminal observation of a
    step that ends in a reset, and the first observation after it."""
    from test_image_oracle import _render, _render_obs
    from mdp_playground_amd import _capi as capi, image_obs, mdp
    cfg = dict(IMG_CFGS[name], seed=9)
    N, T = 192, 12
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    m = mdp.build_mdp(cfg)
    words = env.get_rng_streams(capi.STREAM_IMAGE).copy()
    acts = _img_actions(cfg, (T, N), 4)
    if m.irrelevant:
        # (no integer-observation twin here: without images the reference re-seeds the two state spaces
        # through the Tuple space, so the twin would be a different MDP.)  States after the step come
        # from get_augmented_state(); the terminal observation's state is the terminal pair whose
        # rendering, from the same stream position, equals it.
        tpl = image_obs.build_templates(max(m.S, m.S_irr), m.image)
        n_final = 0
        for t in range(T):
            obs, rew, term, trunc, info = env.step(torch.as_tensor(acts[t], device=env.device))
            obs, fin, d = obs.cpu().numpy(), info["final_obs"].cpu().numpy(), term.cpu().numpy().astype(bool)
            st = env.get_augmented_state()["curr_state"]
            for i in range(N):
                if d[i]:
                    ok = False
                    for s0 in m.terminal_states:
                        for s1 in range(m.S_irr):
                            w2 = words[i].copy()
                            if np.array_equal(_render_obs(m.image, tpl, (int(s0), s1), w2), fin[i]):
                                words[i] = w2
                                ok = True
                                break
                        if ok:
                            break
                    assert ok, (name, t, i)
                    n_final += 1
                assert np.array_equal(_render_obs(m.image, tpl, st[i], words[i]), obs[i]), (name, t, i)
        assert n_final > 50
        assert np.array_equal(words, env.get_rng_streams(capi.STREAM_IMAGE))
        env.close()
        return
    twin_cfg = {k: v for k, v in cfg.items() if not k.startswith("image_")}
    twin = _venv(num_envs=N, autoreset="same_step", **twin_cfg)      # same states, integer observations
    tpl = image_obs.build_templates(m.S, m.image)
    # stream position BEFORE the constructor's reset drew the first observation is not exported;
    # start from the current one: the next draws are those of step 0
    assert np.array_equal(twin._obs.cpu().numpy().shape, (N,))
    n_final = 0
    for t in range(T):
        at = torch.as_tensor(acts[t], device=env.device)
        obs, rew, term, trunc, info = env.step(at)
        sobs, srew, sterm, _, sinfo = twin.step(at)
        assert torch.equal(term, sterm) and torch.equal(rew, srew)
        obs, fin = obs.cpu().numpy(), info["final_obs"].cpu().numpy()
        st, sfin, d = sobs.cpu().numpy(), sinfo["final_obs"].cpu().numpy(), term.cpu().numpy().astype(bool)
        for i in range(N):
            if d[i]:
                assert np.array_equal(_render(m.image, tpl, int(sfin[i]), words[i]), fin[i]), (name, t, i)
                n_final += 1
            assert np.array_equal(_render(m.image, tpl, int(st[i]), words[i]), obs[i]), (name, t, i)
    assert n_final > 50
    assert np.array_equal(words, env.get_rng_streams(capi.STREAM_IMAGE))
    env.close(); twin.close()


@pytest.mark.parametrize("name", ["cfg4", "all100", "irr84"])
def test_image_observations_on_philox_streams_vs_oracle(name):
    """rng="philox" with polygon image observations (VERDICT r2 "missing"): the transforms of tick t come from stream
    (seed, global env id, t, image) in the reference's draw order -- the step's images, then reset()'s where the step ended
    the episode -- and an explicit reset() from its own stream keyed by the reset count.  Against the oracle's Philox
    restatement + Pillow-exact rotation on 160 envs x 12 single steps (observations and terminal observations), then a
    fused rollout == the same steps one by one; env ids offset, so the key is the GLOBAL id."""
    from test_image_oracle import _cfg_struct
    from oracle import oracle as ora
    from mdp_playground_amd import image_obs, mdp
    cfg = dict(IMG_CFGS[name], seed=9)
    N, T, off = 160, 12, 5000
    kw = dict(rng="philox", philox_seed=77, env_id_offset=off)
    env = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    m = mdp.build_mdp(cfg)
    SUB = 2 if m.irrelevant else 1
    tpl = image_obs.build_templates(max(m.S, m.S_irr) if m.irrelevant else m.S, m.image)
    cs = _cfg_struct(m.image, tpl)

    def render(state, xf):
        R, cx, cy, angle, flip = xf
        ri = R - tpl["r_min"]
        tp = tpl["tpl"][state, ri, tpl["cls_x"][state, ri, cx], tpl["cls_y"][state, ri, cy]]
        W, H, half = m.image["width"], m.image["height"], tpl["tpl_size"] // 2
        src = np.zeros((H, W), np.uint8)
        for ty in range(tpl["tpl_size"]):
            y = ty - half + cy
            if 0 <= y < H:
                x0 = cx - half
                lo, hi = max(0, -x0), min(tpl["tpl_size"], W - x0)
                if lo < hi:
                    src[y, x0 + lo:x0 + hi] = tp[ty, lo:hi]
        return ora.image_rotate_flip_transpose(src, angle, flip)[:, :, None]

    def render_obs(states, xfs):
        return np.concatenate([render(int(s_), xf) for s_, xf in zip(np.atleast_1d(states), xfs)], axis=0)

    # the constructor's reset(): explicit-reset stream (id 11), reset count 0
    st0 = env.get_augmented_state()["curr_state"]
    first = env._obs.cpu().numpy()
    for i in range(0, N, 7):
        xfs = ora.image_draw_philox(cs, 77, off + i, 0, 11, SUB)
        assert np.array_equal(render_obs(st0[i], xfs), first[i]), (name, i)
    acts = _img_actions(cfg, (T + 20, N), 4)
    b = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    n_final = 0
    for t in range(T):
        obs, rew, term, trunc, info = env.step(torch.as_tensor(acts[t], device=env.device))
        obs, fin, d = obs.cpu().numpy(), info["final_obs"].cpu().numpy(), term.cpu().numpy().astype(bool)
        st = env.get_augmented_state()["curr_state"]
        for i in range(0, N, 7):
            xfs = ora.image_draw_philox(cs, 77, off + i, t, 2, 2 * SUB if d[i] else SUB)
            if d[i]:
                # the terminal observation's state is not exported: it is the terminal state (pair) whose rendering
                # with the first SUB transforms equals final_obs
                cands = [(s0,) for s0 in m.terminal_states] if SUB == 1 else \
                        [(s0, s1) for s0 in m.terminal_states for s1 in range(m.S_irr)]
                assert any(np.array_equal(render_obs(c, xfs[:SUB]), fin[i]) for c in cands), (name, t, i)
                n_final += 1
                assert np.array_equal(render_obs(st[i], xfs[SUB:]), obs[i]), (name, t, i)
            else:
                assert np.array_equal(render_obs(st[i], xfs), obs[i]), (name, t, i)
    assert n_final > 10
    # a fused rollout (batches of 16 steps) == single steps, on twin envs
    o1 = env.rollout(torch.as_tensor(acts[T:], device=env.device))
    for t in range(T):
        b.step(torch.as_tensor(acts[t], device=b.device))
    for k in range(20):
        o, r, te, tr, _ = b.step(torch.as_tensor(acts[T + k], device=b.device))
        assert torch.equal(o, o1[0][k]) and torch.equal(r, o1[1][k]) and torch.equal(te, o1[2][k]), (name, k)
    env.close(); b.close()


def test_cfg4_at_bench_size_fast_vs_general_renderer_and_oracle():
    """BASELINE configs[3] at ITS size: 8 192 envs x 80 fused steps = two pipelined batches (64 + 16).  The
    persistent fast renderer's grid, the slots it leaves free and the counters its waves claim their images from depend on
    the batch size, so the size it is benchmarked at is the size it is checked at: every pixel of every
    image against the general renderer (k_image_obs, NO_IMGFAST), the pipeline against the unpipelined launch order
    (NO_IMG_OVERLAP), and a strided sample of envs against the oracle's draw + Pillow-exact rotation."""
    import bench
    from test_image_oracle import _render
    from mdp_playground_amd import _capi as capi, image_obs, mdp
    wl = bench.WORKLOADS["cfg4"]
    cfg, N, K = wl["config"], wl["envs"], 80
    assert N == 8192
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    c = _venv(num_envs=N, autoreset="same_step", **cfg)
    d = _venv(num_envs=N, autoreset="same_step", **cfg)
    b.set_kernel_options("NO_IMGFAST")
    c.set_kernel_options("NO_IMG_OVERLAP")
    d.set_kernel_options("NO_IMG_NEARTAB")              # (the fast renderer walking the bounding box instead of the near-dword table)
    assert a.rollout_kernel_name(K).startswith("k_image_obs_fast<") and b.rollout_kernel_name(K) == "k_image_obs"
    twin = _venv(num_envs=N, autoreset="same_step", **{k: v for k, v in cfg.items() if not k.startswith("image_")})
    words0 = a.get_rng_streams(capi.STREAM_IMAGE).copy()
    acts = bench.make_actions(wl, K, N, a.device, 77)
    oa, ra, ta, _ = a.rollout(acts)
    ob, rb, tb, _ = b.rollout(acts)
    oc, rc, tc, _ = c.rollout(acts)
    od, rd, td, _ = d.rollout(acts)
    st, rs, ts, _ = twin.rollout(acts)
    torch.cuda.synchronize()
    assert tuple(oa.shape) == (K, N, 84, 84, 1)
    for k in range(K):                                    # (row by row: a whole-tensor compare allocates another 1.8 GB)
        assert torch.equal(oa[k], ob[k]), k
        assert torch.equal(oa[k], oc[k]), k
        assert torch.equal(oa[k], od[k]), k
    assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(ra, rc) and torch.equal(ta, tc)
    assert torch.equal(ra, rs) and torch.equal(ta, ts) and ta.any() and not ta.all()
    for s_ in (capi.STREAM_IMAGE, capi.STREAM_ENV):
        assert np.array_equal(a.get_rng_streams(s_), b.get_rng_streams(s_))
        assert np.array_equal(a.get_rng_streams(s_), c.get_rng_streams(s_))
    # the oracle on every 257th env: the state sequence from the integer-observation twin; a step that ends in a
    # reset draws the terminal observation first (not rendered by a rollout, and with shift + rotate its draw count
    # does not depend on the state), then the first observation of the next episode
    m = mdp.build_mdp(cfg)
    tpl = image_obs.build_templates(m.S, m.image)
    st_h, term_h = st.cpu().numpy(), ta.cpu().numpy().astype(bool)
    end = a.get_rng_streams(capi.STREAM_IMAGE)
    for i in range(3, N, 257):
        w = words0[i].copy()
        got = oa[:, i].cpu().numpy()
        for k in range(K):
            if term_h[k, i]:
                _render(m.image, tpl, 0, w)
            assert np.array_equal(_render(m.image, tpl, int(st_h[k, i]), w), got[k]), (i, k)
        assert np.array_equal(w, end[i]), i
    assert torch.equal(ra, rd) and torch.equal(ta, td)
    for e in (a, b, c, d, twin):
        e.close()


# ----------------------------------------------------------------------------- ImageContinuous
@pytest.mark.parametrize("name", gu.IMAGE_CONT)
def test_continuous_image_observations_vs_reference_golden(name):
    """Continuous envs with image_representations: every pixel of the RGB pictures the reference
    drew (Pillow rectangles and ellipses), rewards and flags under the every-step
    clip-and-zero-derivatives quirk, masked resets, terminal pictures of reset steps."""
    g = gu.load(name)
    E, T, D = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_obs"])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t], device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy(), g["obs"][:, t]), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra])
    st = env.get_augmented_state()
    assert np.array_equal(st["state_derivatives"].view(np.uint32), g["sd"][:, -1].view(np.uint32)) or g["reset_after"][:, -1].any()
    env.close()


def test_continuous_image_batch_vs_oracle_and_fused():
    """512 envs x 40 steps of a 4-D env with terminal hypercubes and both noises: the fused rollout
    (batches of 16 steps: state kernel + render kernel) equals single steps bit for bit, and every
    4th env's pictures (observation, terminal observation of reset steps, first observation after)
    equal the oracle's."""
    from oracle import oracle as ora
    cfg = dict(state_space_type="continuous", state_space_dim=4, relevant_indices=[0, 1],
               transition_dynamics_order=2, inertia=1.0, time_unit=1.0, state_space_max=4, action_space_max=1,
               make_denser=True, target_point=[1.5, -2.0], target_radius=0.7,
               terminal_states=[[-2.0, 2.0], [3.0, 0.0]], term_state_edge=1.5, transition_noise=0.1,
               reward_noise=0.05, reward_function="move_to_a_point", image_representations=True,
               image_width=64, image_height=80, seed=4)
    N, K = 512, 40
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    assert a.rollout_kernel_name(K).startswith("k_imagec_obs<")
    acts = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, size=(K, N, 4)).astype(np.float32), device=a.device)
    obs, rew, term, trunc = a.rollout(acts)
    assert term.any()
    fins = []
    for t in range(K):
        o, r, te, tr, info = b.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(te, term[t]), t
        fins.append(info["final_obs"].cpu().numpy().copy())
    obs_h, term_h, rew_h, acts_h = obs.cpu().numpy(), term.cpu().numpy(), rew.cpu().numpy(), acts.cpu().numpy()
    m = a.mdps[0]

    def picture(state):
        return ora.image_continuous_render(64, 80, 5, state, 4.0, cfg["target_point"], m.box_lo, m.box_hi)
    n_final = 0
    for i in range(0, N, 4):
        o = _oracle_for(a, i)
        o.set_image_quirk(True)
        o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i])
        o.reset()
        for t in range(K):
            st, r, is32, d = o.step(acts_h[t, i])
            assert d == bool(term_h[t, i]) and np.float32(r) == rew_h[t, i], (i, t)
            if d:
                assert np.array_equal(picture(st), fins[t][i]), (i, t)
                n_final += 1
                st = o.reset()
            assert np.array_equal(picture(st), obs_h[t, i]), (i, t)
    assert n_final > 20
    a.close(); b.close()


@pytest.mark.parametrize("name", gu.IMAGE_GRID)
def test_grid_image_observations_vs_reference_golden(name):
    """Grid envs with image_representations: every pixel of the reference's pictures (grid lines,
    terminal cells, discs at cell centres, irrelevant grid), rewards, flags, masked resets."""
    g = gu.load(name)
    E, T, G = g["action"].shape
    env = _venv(autoreset="disabled", **_seeds_or_cfg(name))
    assert np.array_equal(env._obs.cpu().numpy(), g["init_obs"])
    for t in range(T):
        a = torch.as_tensor(g["action"][:, t].astype(np.int32), device=env.device)
        obs, rew, term, trunc, _ = env.step(a)
        assert np.array_equal(obs.cpu().numpy(), g["obs"][:, t]), (name, t)
        assert np.array_equal(term.cpu().numpy(), g["done"][:, t]), (name, t)
        assert np.array_equal(rew.cpu().numpy(), g["reward"][:, t].astype(np.float32)), (name, t)
        ra = g["reset_after"][:, t]
        if ra.any():
            o, _ = env.reset(mask=torch.as_tensor(ra, device=env.device))
            assert np.array_equal(o.cpu().numpy()[ra], g["reset_obs"][:, t][ra])
    assert np.array_equal(env.get_augmented_state()["curr_state"], g["curr_state"][:, -1]) or g["reset_after"][:, -1].any()
    env.close()


def test_grid_image_fused_equals_single_steps_and_oracle():
    from oracle import oracle as ora
    cfg = dict(state_space_type="grid", grid_shape=(6, 5), reward_function="move_to_a_point", make_denser=True,
               target_point=[2, 3], irrelevant_features=True, transition_noise=0.2, terminal_states=[[0, 0], [5, 4]],
               image_representations=True, image_width=48, image_height=64, seed=8)
    N, K = 384, 40
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    r = np.random.default_rng(6)
    acts = np.zeros((K, N, 4), np.int32)
    np.put_along_axis(acts, r.integers(0, 4, size=(K, N, 1)), r.integers(-1, 2, size=(K, N, 1)).astype(np.int32), axis=2)
    acts_t = torch.as_tensor(acts, device=a.device)
    obs, rew, term, trunc = a.rollout(acts_t)
    assert term.any()
    for t in range(K):
        o, rr, te, tr, info = b.step(acts_t[t])
        assert torch.equal(o, obs[t]) and torch.equal(rr, rew[t]) and torch.equal(te, term[t]), t
    obs_h, term_h = obs.cpu().numpy(), term.cpu().numpy()
    m = a.mdps[0]
    for i in range(0, N, 6):
        o = _oracle_for(a, i)
        o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i], a.seeded_streams[4][i])
        o.reset()
        for t in range(K):
            st, r_, d = o.step(acts[t, i])
            assert d == bool(term_h[t, i]), (i, t)
            if d:
                st = o.reset()
            pic = ora.image_grid_render(48, 64, 5, m.grid_shape, st, cfg["target_point"], cfg["terminal_states"])
            assert np.array_equal(pic, obs_h[t, i]), (i, t)
    a.close(); b.close()


@pytest.mark.parametrize("rng", ["numpy", "philox"])
@pytest.mark.parametrize("name", ["cfg4", "all100", "irr84"])
def test_polygon_images_with_next_step_autoreset_vs_oracle(name, rng):
    """gymnasium's next-step autoreset with polygon image observations: the call after an episode's last step is reset()
    alone -- it draws ONE observation's transforms from the image stream like any step (reset() -> get_image_representation,
    rl_toy_env.py:2237; image_multi_discrete.py:272-288), reward 0, flags False, no terminal observation.  160 envs x 14
    single steps against the oracle's draw + Pillow-exact rotation, then a fused rollout == the same steps one by one."""
    from test_image_oracle import _render_obs, _cfg_struct
    from oracle import oracle as ora
    from mdp_playground_amd import _capi as capi, image_obs, mdp
    cfg = dict(IMG_CFGS[name], seed=9)
    N, T = 160, 14
    kw = dict(rng="philox", philox_seed=31, env_id_offset=700) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="next_step", **kw, **cfg)
    b = _venv(num_envs=N, autoreset="next_step", **kw, **cfg)
    m = mdp.build_mdp(cfg)
    SUB = 2 if m.irrelevant else 1
    tpl = image_obs.build_templates(max(m.S, m.S_irr) if m.irrelevant else m.S, m.image)
    words = env.get_rng_streams(capi.STREAM_IMAGE).copy() if rng == "numpy" else None
    acts = _img_actions(cfg, (T + 20, N), 4)
    pending = np.zeros(N, bool)
    n_reset = 0
    if rng == "philox":
        cs = _cfg_struct(m.image, tpl)
        W, H, ts = m.image["width"], m.image["height"], tpl["tpl_size"]

        def render(state, xf):
            R, cx, cy, angle, flip = xf
            ri = R - tpl["r_min"]
            tp = tpl["tpl"][state, ri, tpl["cls_x"][state, ri, cx], tpl["cls_y"][state, ri, cy]]
            src = np.zeros((H, W), np.uint8)
            for ty in range(ts):
                y = ty - ts // 2 + cy
                x0 = cx - ts // 2
                lo, hi = max(0, -x0), min(ts, W - x0)
                if 0 <= y < H and lo < hi:
                    src[y, x0 + lo:x0 + hi] = tp[ty, lo:hi]
            return ora.image_rotate_flip_transpose(src, angle, flip)[:, :, None]
    for t in range(T):
        obs, rew, term, trunc, info = env.step(torch.as_tensor(acts[t], device=env.device))
        assert "final_obs" not in info
        obs, rew, d = obs.cpu().numpy(), rew.cpu().numpy(), (term | trunc).cpu().numpy().astype(bool)
        assert not d[pending].any() and not rew[pending].any()             # a reset call: reward 0, flags False
        n_reset += int(pending.sum())
        st = env.get_augmented_state()["curr_state"]
        for i in range(0, N, 3 if rng == "numpy" else 7):
            if rng == "numpy":
                assert np.array_equal(_render_obs(m.image, tpl, st[i], words[i]), obs[i]), (name, t, i)
            else:
                xfs = ora.image_draw_philox(cs, 31, 700 + i, t, 2, SUB)
                pic = np.concatenate([render(int(s_), xf) for s_, xf in zip(np.atleast_1d(st[i]), xfs)], axis=0)
                assert np.array_equal(pic, obs[i]), (name, t, i)
        pending = d
    assert n_reset > 10
    if rng == "numpy":
        got = env.get_rng_streams(capi.STREAM_IMAGE)
        assert np.array_equal(words[::3], got[::3])
    # fused (batches of 16 steps) == single steps
    for t in range(T):
        b.step(torch.as_tensor(acts[t], device=b.device))
    o1, r1, t1, _ = env.rollout(torch.as_tensor(acts[T:], device=env.device))
    for k in range(20):
        o, r, te, tr, _ = b.step(torch.as_tensor(acts[T + k], device=b.device))
        assert torch.equal(o, o1[k]) and torch.equal(r, r1[k]) and torch.equal(te, t1[k]), (name, k)
    env.close(); b.close()


def test_continuous_and_grid_images_with_next_step_autoreset_vs_oracle():
    """The same mode for the continuous and the grid pictures (no image stream there): the picture of a reset call is the
    new episode's start state's."""
    from oracle import oracle as ora
    cfg = dict(state_space_type="continuous", state_space_dim=4, relevant_indices=[0, 1],
               transition_dynamics_order=1, inertia=1.0, time_unit=1.0, state_space_max=4, action_space_max=1,
               make_denser=True, target_point=[1.5, -2.0], target_radius=0.7,
               terminal_states=[[-2.0, 2.0], [3.0, 0.0]], term_state_edge=1.5, transition_noise=0.1,
               reward_function="move_to_a_point", image_representations=True, image_width=64, image_height=80, seed=4)
    N, K = 256, 40
    a = _venv(num_envs=N, autoreset="next_step", **cfg)
    acts = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, size=(K, N, 4)).astype(np.float32), device=a.device)
    obs, rew, term, trunc = a.rollout(acts)
    obs_h, term_h, rew_h, acts_h = obs.cpu().numpy(), term.cpu().numpy(), rew.cpu().numpy(), acts.cpu().numpy()
    m = a.mdps[0]
    n_reset = 0
    for i in range(0, N, 4):
        o = _oracle_for(a, i)
        o.set_image_quirk(True)
        o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i])
        o.reset()
        pend = False
        for t in range(K):
            if pend:
                st, r, d = o.reset(), 0.0, False
                n_reset += 1
            else:
                st, r, is32, d = o.step(acts_h[t, i])
            assert d == bool(term_h[t, i]) and np.float32(r) == rew_h[t, i], (i, t)
            pic = ora.image_continuous_render(64, 80, 5, st, 4.0, cfg["target_point"], m.box_lo, m.box_hi)
            assert np.array_equal(pic, obs_h[t, i]), (i, t)
            pend = d
    assert n_reset > 10
    a.close()
    cfg = dict(state_space_type="grid", grid_shape=(6, 5), reward_function="move_to_a_point", make_denser=True,
               target_point=[2, 3], irrelevant_features=True, transition_noise=0.2, terminal_states=[[0, 0], [5, 4]],
               image_representations=True, image_width=48, image_height=64, seed=8)
    N, K = 192, 40
    a = _venv(num_envs=N, autoreset="next_step", **cfg)
    r = np.random.default_rng(6)
    acts = np.zeros((K, N, 4), np.int32)
    np.put_along_axis(acts, r.integers(0, 4, size=(K, N, 1)), r.integers(-1, 2, size=(K, N, 1)).astype(np.int32), axis=2)
    obs, rew, term, trunc = a.rollout(torch.as_tensor(acts, device=a.device))
    obs_h, term_h = obs.cpu().numpy(), term.cpu().numpy()
    m = a.mdps[0]
    n_reset = 0
    for i in range(0, N, 6):
        o = _oracle_for(a, i)
        o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i], a.seeded_streams[4][i])
        o.reset()
        pend = False
        for t in range(K):
            if pend:
                st, d = o.reset(), False
                n_reset += 1
            else:
                st, r_, d = o.step(acts[t, i])
            assert d == bool(term_h[t, i]), (i, t)
            pic = ora.image_grid_render(48, 64, 5, m.grid_shape, st, cfg["target_point"], cfg["terminal_states"])
            assert np.array_equal(pic, obs_h[t, i]), (i, t)
            pend = d
    assert n_reset > 5
    a.close()


# ----------------------------------------------------------------------------- BASELINE full sizes
def _cfg(name, seed):
    return dict(gu.CASES[name]["config"], seed=seed)


@pytest.mark.parametrize("name,N", [("d_cfg2", 65536), ("c_cfg3", 65536), ("c_cfg5", 65536),
                                    ("d_irr_noise", 65536), ("g_noise_sparse", 65536)])
def test_full_size_rollout_equals_single_steps_and_oracle_sample(name, N):
    """At BASELINE's 65 536 envs the oracle cannot replay everything in seconds, so check
    size-independent properties: (1) one fused K-step launch == K single-step launches bit for
    bit (two different launch shapes of the same arithmetic), (2) a strided sample of envs ==
    the oracle, (3) state export -> import -> continue reproduces the continuation."""
    cfg = _cfg(name, 17)
    T = 64                       # >= 32: the multi-wave (pipelined / helper-wave) kernels serve the fused launch
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    if name == "d_cfg2":         # the benchmarked kernel itself is what the oracle sees here
        assert a.rollout_kernel_name(T).startswith("k_discrete_rollout_lean<"), a.rollout_kernel_name(T)
        assert a.rollout_kernel_name(1).startswith("k_discrete_step1<")       # (round 5: mdpp_step has its own kernel)
    if name in ("c_cfg3", "c_cfg5"):
        assert a.rollout_kernel_name(1).startswith("k_continuous_step1<"), a.rollout_kernel_name(1)
    rng = np.random.default_rng(0)
    from mdp_playground_amd import _capi as capi
    streams = [0, 1]
    if a.kind == "discrete" and a._irr:
        m = a.mdps[0]
        acts = torch.as_tensor(np.stack([rng.integers(0, m.A, size=(T, N)), rng.integers(0, m.A_irr, size=(T, N))],
                                        axis=2).astype(np.int32), device=a.device)
        streams.append(capi.STREAM_SPACE_IRR)
    elif a.kind == "discrete":
        acts = torch.as_tensor(rng.integers(0, 8, size=(T, N)).astype(np.int32), device=a.device)
    elif a.kind == "grid":
        G = len(a.mdps[0].grid_shape)
        ac = np.zeros((T, N, G), np.int32)
        np.put_along_axis(ac, rng.integers(0, G, size=(T, N, 1)), rng.integers(-1, 2, size=(T, N, 1)).astype(np.int32), axis=2)
        acts = torch.as_tensor(ac, device=a.device)
        streams.append(capi.STREAM_ACTION)
    else:
        acts = torch.as_tensor(rng.uniform(-1, 1, size=(T, N, 12)).astype(np.float32), device=a.device)
    init = a._obs.cpu().numpy().copy()
    obs, rew, term, trunc = a.rollout(acts)
    for t in range(T):
        o, r, te, tr, _ = b.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(te, term[t]), (name, t)
    # (2) oracle on a sample
    obs_h, rew_h, term_h = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    acts_h = acts.cpu().numpy()
    for i in range(0, N, 4099):
        o = _oracle_for(a, i)
        if a.kind == "grid":
            o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i], a.seeded_streams[capi.STREAM_ACTION][i])
        else:
            o.set_rng(a.seeded_streams[0][i], a.seeded_streams[1][i])
            if a.kind == "discrete" and a._irr:
                o.set_rng_irr(a.seeded_streams[capi.STREAM_SPACE_IRR][i])
        s0 = o.reset()
        assert np.array_equal(np.asarray(s0), init[i])
        eo, er, ed, ero = o.rollout(acts_h[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(obs_h[:, i], eo) and np.array_equal(term_h[:, i], ed), (name, i)
        assert np.array_equal(rew_h[:, i], er.astype(np.float32)), (name, i)
    # (3) checkpoint round trip: b's state + streams into a fresh env, both continue identically
    c = _venv(num_envs=N, autoreset="same_step", **cfg)
    c.set_augmented_state(b.get_augmented_state())
    for s in streams:
        c._put_stream(s, b.get_rng_streams(s))
    o1, r1, t1, _ = b.rollout(acts[:8])
    o2, r2, t2, _ = c.rollout(acts[:8])
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2)
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("kernel", ["lean", "pipe"])
def test_bench_shape_kernel_vs_oracle_every_env(kernel):
    """Exactly what `python bench.py --gpus 1 --steps 20 --warmup 5` launches (bench.WORKLOADS["cfg2"]: 65 536
    envs, seed 0, fused launches of 512 steps, bench.make_actions(seed 12345)): k_discrete_rollout_lean (and
    k_discrete_rollout_pipe, which serves the shapes lean does not, selected with NO_LEAN)
    against the oracle on EVERY env for the first launch, on every 16th env for the second launch
    (state and streams carried across launches), and the env streams' end states."""
    import bench
    wl = bench.WORKLOADS["cfg2"]
    N, F = wl["envs"], 512
    env = _venv(num_envs=N, autoreset="same_step", **wl["config"])
    if kernel == "pipe":
        env.set_kernel_options("NO_LEAN")
    assert env.rollout_kernel_name(F).startswith(f"k_discrete_rollout_{kernel}<"), env.rollout_kernel_name(F)
    acts = bench.make_actions(wl, F, N, env.device, 12345)
    init = env._obs.cpu().numpy().copy()
    res = []
    for _ in range(2):
        obs, rew, term, trunc = env.rollout(acts)
        assert not trunc.any()
        res.append((obs.cpu().numpy().T.copy(), rew.cpu().numpy().T.copy(), term.cpu().numpy().T.copy()))
    end = env.get_rng_streams(0)
    acts_t = acts.cpu().numpy().T.copy()                     # [N, F]
    from oracle import oracle as ora
    m = env.mdps[0]
    rt = m.reward_table()
    for i in range(N):
        o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P, rt, m.terminal_states,
                               m.init_dist, m.transition_noise, m.reward_noise, m.reward_scale, m.reward_shift,
                               m.term_state_reward)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert o.reset() == int(init[i])
        for launch in range(2 if i % 16 == 0 else 1):
            eo, er, ed, ero = o.rollout(acts_t[i], None)
            eo[ed] = ero[ed]
            obs_h, rew_h, term_h = res[launch]
            assert np.array_equal(obs_h[i], eo) and np.array_equal(term_h[i], ed), (i, launch)
            assert np.array_equal(rew_h[i], er.astype(np.float32)), (i, launch)
        if i % 16 == 0:
            assert np.array_equal(o.get_rng()[0][:4], end[i][:4]), i
    assert (env.status() == 0).all()
    env.close()


@pytest.mark.timeout(1500)
def test_bench_shape_cfg5_kernel_vs_oracle_every_env():
    """the cfg5 leg of the bench line (bench.WORKLOADS["cfg5"]: 65 536 envs, fused launches of 512 steps,
    bench.make_actions(seed 12345), numpy-exact streams) is the kernel whose stream consumption is data dependent --
    generator, walker and consumer waves (k_continuous_rollout_fast<..., NPROD=2>), lanes drifting apart by what their
    ziggurat rejections consumed.  Against the oracle on EVERY env for one launch: float32 states bit for bit, rewards,
    flags, and the end states of both streams of every env (the generator un-draws what the walker did not take); a
    second launch (state and streams carried over) on every 64th env."""
    import bench
    wl = bench.WORKLOADS["cfg5"]
    N, F = wl["envs"], 512
    env = _venv(num_envs=N, autoreset="same_step", **wl["config"])
    kname = env.rollout_kernel_name(F)
    assert kname.startswith("k_continuous_rollout_fast<") and "PHILOX=0" in kname and "NPROD=2" in kname, kname
    acts = bench.make_actions(wl, F, N, env.device, 12345)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(acts)
    assert not trunc.any()
    end1 = (env.get_rng_streams(0), env.get_rng_streams(1))
    obs2, rew2, term2, _ = env.rollout(acts)
    end2 = (env.get_rng_streams(0), env.get_rng_streams(1))
    assert (env.status() == 0).all()
    B = 2048                                    # envs per host block (actions + observations: 100 MB a block)
    for b0 in range(0, N, B):
        a_h = acts[:, b0:b0 + B].cpu().numpy()
        o_h, r_h, t_h = obs[:, b0:b0 + B].cpu().numpy(), rew[:, b0:b0 + B].cpu().numpy(), term[:, b0:b0 + B].cpu().numpy()
        o2_h, r2_h, t2_h = obs2[:, b0:b0 + B].cpu().numpy(), rew2[:, b0:b0 + B].cpu().numpy(), term2[:, b0:b0 + B].cpu().numpy()
        for j in range(B):
            i = b0 + j
            o = _oracle_for(env, i)
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            assert np.array_equal(o.reset(), init[i]), i
            a_i = np.ascontiguousarray(a_h[:, j])
            eo, er, ed, ero = o.rollout(a_i, None)
            eo[ed] = ero[ed]
            assert np.array_equal(o_h[:, j].view(np.uint32), eo.view(np.uint32)), i
            assert np.array_equal(t_h[:, j], ed) and np.array_equal(r_h[:, j], er.astype(np.float32)), i
            ge, gs = o.get_rng()
            assert np.array_equal(ge[:4], end1[0][i][:4]) and np.array_equal(gs[:4], end1[1][i][:4]), i
            if i % 64 == 0:
                eo, er, ed, ero = o.rollout(a_i, None)
                eo[ed] = ero[ed]
                assert np.array_equal(o2_h[:, j].view(np.uint32), eo.view(np.uint32)), i
                assert np.array_equal(t2_h[:, j], ed) and np.array_equal(r2_h[:, j], er.astype(np.float32)), i
                ge, gs = o.get_rng()
                assert np.array_equal(ge[:4], end2[0][i][:4]) and np.array_equal(gs[:4], end2[1][i][:4]), i
    env.close()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("workload", ["d_s8_rn0", "d_s50_rn0", "c_d2_n0"])
def test_sigma_zero_bench_shapes_vs_oracle_every_env(workload):
    """Round 6: the reference's commonest experiment shapes -- noise keys present with sigma 0 (bench.WORKLOADS d_s8_rn0,
    d_s50_rn0, c_d2_n0) -- on the advance-only draws of the default dispatch (lean Z0 / quiet's skip draw / the walker
    without a normals ring), against the ORACLE (which draws every normal the way numpy does) on EVERY env of the bench
    size for one fused launch of 512 steps: observations, rewards and flags bit for bit, and the end state of both
    streams of every env -- i.e. every draw consumed exactly the words numpy's ziggurat consumes."""
    import bench
    wl = bench.WORKLOADS[workload]
    N, F = wl["envs"], 512
    env = _venv(num_envs=N, autoreset="same_step", **wl["config"])
    kname = env.rollout_kernel_name(F)
    assert kname.startswith({"d_s8_rn0": "k_discrete_rollout_lean<", "d_s50_rn0": "k_discrete_rollout_quiet<",
                             "c_d2_n0": "k_continuous_rollout_fast<"}[workload]), kname
    assert ("Z0=1" in kname) == (workload != "d_s50_rn0") and ("ROLES=3" in kname) == (workload == "d_s50_rn0"), kname
    acts = bench.make_actions(wl, F, N, env.device, 12345)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(acts)
    assert not trunc.any() and (env.status() == 0).all()
    end = (env.get_rng_streams(0), env.get_rng_streams(1))
    if wl["kind"] == "discrete":
        from oracle import oracle as ora
        m = env.mdps[0]
        rt = m.reward_table()
        obs_h, rew_h, term_h = obs.cpu().numpy().T.copy(), rew.cpu().numpy().T.copy(), term.cpu().numpy().T.copy()
        acts_t = acts.cpu().numpy().T.copy()
        for i in range(N):
            o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P, rt, m.terminal_states,
                                   m.init_dist, m.transition_noise, m.reward_noise, m.reward_scale, m.reward_shift, m.term_state_reward)
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            assert o.reset() == int(init[i])
            eo, er, ed, ero = o.rollout(acts_t[i], None)
            eo[ed] = ero[ed]
            assert np.array_equal(obs_h[i], eo) and np.array_equal(term_h[i], ed), i
            assert np.array_equal(rew_h[i].view(np.uint32), er.astype(np.float32).view(np.uint32)), i
            ge, gs = o.get_rng()
            assert np.array_equal(ge[:4], end[0][i][:4]) and np.array_equal(gs[:4], end[1][i][:4]), i
    else:
        B = 4096
        for b0 in range(0, N, B):
            a_h = acts[:, b0:b0 + B].cpu().numpy()
            o_h, r_h, t_h = obs[:, b0:b0 + B].cpu().numpy(), rew[:, b0:b0 + B].cpu().numpy(), term[:, b0:b0 + B].cpu().numpy()
            for j in range(B):
                i = b0 + j
                o = _oracle_for(env, i)
                o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
                assert np.array_equal(o.reset(), init[i]), i
                eo, er, ed, ero = o.rollout(np.ascontiguousarray(a_h[:, j]), None)
                eo[ed] = ero[ed]
                assert np.array_equal(o_h[:, j].view(np.uint32), eo.view(np.uint32)), i
                assert np.array_equal(t_h[:, j], ed) and np.array_equal(r_h[:, j].view(np.uint32), er.astype(np.float32).view(np.uint32)), i
                ge, gs = o.get_rng()
                assert np.array_equal(ge[:4], end[0][i][:4]) and np.array_equal(gs[:4], end[1][i][:4]), i
    env.close()


@pytest.mark.parametrize("variant", ["l3_delay4", "sf_l1", "rn_l2", "sf_rn0", "noreset_rn", "s16_l1"])
def test_per_env_mdps_on_the_role_split_kernel_vs_oracle(variant):
    """Round 6: one MDP PER ENV (seeds=[...]: env i is the reference's RLToyEnv(seed=i) -- what every golden
    uses) on k_discrete_rollout_quiet<PE=1>: each lane's tables in its slot of the workgroup's LDS, E / O / H (or X) roles.
    1 024 different MDPs; fused rollouts (the role-split kernel), short ones and single steps (the general kernel, same
    handle: the two hand the state to each other); EVERY env against its own oracle -- observations, rewards, flags and the
    end states of both streams after every launch."""
    from mdp_playground_amd import _capi as capi
    base = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8)
    cfg, kw = {"l3_delay4": (dict(base, delay=4, sequence_length=3), {}),
               "sf_l1": (dict(base, delay=1, sequence_length=1, reward_density=0.25, terminal_state_density=0.25), {}),
               "rn_l2": (dict(base, delay=2, sequence_length=2, reward_noise=0.5, reward_scale=2.0, term_state_reward=-1.0),
                         dict(max_episode_steps=13)),
               "sf_rn0": (dict(base, delay=0, sequence_length=1, reward_noise=0.0, reward_density=0.25, terminal_state_density=0.25), {}),
               "noreset_rn": (dict(base, delay=1, sequence_length=2, reward_noise=0.3), dict(autoreset="disabled")),
               "s16_l1": (dict(base, state_space_size=16, action_space_size=12, delay=0, sequence_length=1, reward_density=0.2,
                               terminal_state_density=0.125), {})}[variant]
    N = 1024
    kwargs = dict(autoreset="same_step")
    kwargs.update(kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = _venv(seeds=list(range(100, 100 + N)), **kwargs, **cfg)
    kn = env.rollout_kernel_name(64)
    if variant in ("rn_l2", "noreset_rn"):       # (values formed: the X wave's 32 KiB values ring does not fit beside 256 table slots -> the general kernel)
        assert kn.startswith("k_discrete_step<"), kn
    else:
        assert kn.startswith("k_discrete_rollout_quiet<") and "PE=1" in kn and ("SF=1" in kn) == (variant in ("sf_l1", "sf_rn0", "s16_l1")), kn
    assert env.rollout_kernel_name(8).startswith("k_discrete_step<")
    auto = kwargs["autoreset"] == "same_step"
    horizon = kwargs.get("max_episode_steps", 0)
    A = env.mdps[0].A
    r = np.random.default_rng(7)
    init = env._obs.cpu().numpy().copy()
    oracles = []
    for i in range(N):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(np.asarray(o.reset()), init[i])
        oracles.append([o, 0])
    for K in (64, 1, 40, 8, 96):
        acts = r.integers(0, A, size=(K, N)).astype(np.int32)
        at = torch.as_tensor(acts, device=env.device)
        if K == 1:
            o1, r1, t1, tr1, _ = env.step(at[0])
            obs, rew, term, trunc = (x[None].cpu().numpy() for x in (o1, r1, t1, tr1))
        else:
            obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(at))
        env_end, sp_end = env.get_rng_streams(capi.STREAM_ENV), env.get_rng_streams(capi.STREAM_SPACE)
        for i, rec in enumerate(oracles):
            o = rec[0]
            for t in range(K):
                st, rr, d = o.step(acts[t, i])
                rec[1] += 1
                tr = bool(horizon) and rec[1] >= horizon
                assert d == bool(term[t, i]) and tr == bool(trunc[t, i]), (variant, K, i, t)
                assert np.float32(rr) == rew[t, i], (variant, K, i, t, rr, rew[t, i])
                if auto and (d or tr):
                    st = o.reset(explicit=False)
                    rec[1] = 0
                assert np.array_equal(obs[t, i], np.asarray(st)), (variant, K, i, t)
            assert np.array_equal(o.get_rng()[0][:4], env_end[i][:4]), (variant, K, i)
            assert np.array_equal(o.get_rng()[1][:4], sp_end[i][:4]), (variant, K, i)
    assert int(env.status().sum()) == 0
    env.close()


def test_edge_shapes_and_errors():
    cfg = _cfg("d_cfg2", 1)
    # ragged sizes: 1 env, and a batch that is not a multiple of the wave or block size
    for N in (1, 63, 257):
        env = _venv(num_envs=N, autoreset="same_step", **cfg)
        acts = torch.randint(0, 8, (5, N), device=env.device, dtype=torch.int32)
        obs, rew, term, trunc = env.rollout(acts)
        assert obs.shape == (5, N) and int(obs.max()) < 8
        env.close()
    env = _venv(num_envs=64, autoreset="disabled", **cfg)
    with pytest.raises(ValueError):
        env.step(torch.zeros(65, dtype=torch.int32, device=env.device))       # wrong batch size
    # out-of-range discrete action: the reference raises IndexError per env; here it is flagged
    a = torch.zeros(64, dtype=torch.int32, device=env.device)
    a[3] = 8
    a[5] = -1            # numpy negative indexing is legal in the reference: last action
    env.step(a)
    st = env.status()
    assert st[3] == 1 and st.sum() == 1
    assert env.status().sum() == 0                                              # cleared by the read
    env.close()
    # continuous: wrong dtype is an error, out-of-box action means "stay" (:1671) + status bit
    ccfg = _cfg("c_cfg3", 2)
    env = _venv(num_envs=32, autoreset="disabled", **ccfg)
    with pytest.raises(TypeError):
        env.step(torch.zeros((32, 12), dtype=torch.float64, device=env.device))
    before = env._obs.clone()
    a = torch.zeros((32, 12), dtype=torch.float32, device=env.device)
    a[7, 2] = 1.5
    obs, *_ = env.step(a)
    assert torch.equal(obs[7], before[7]) and env.status()[7] == 1
    env.close()
    # truncation (RLToyFiniteHorizon-v0 semantics)
    from mdp_playground_amd import make_vec
    env = make_vec("RLToyVecFiniteHorizon-v0", num_envs=16, autoreset="same_step", max_episode_steps=3,
                   **dict(cfg, terminal_state_density=0.0))
    flags = []
    for _ in range(7):
        _, _, te, tr, _ = env.step(torch.zeros(16, dtype=torch.int32, device=env.device))
        flags.append((bool(te.any()), bool(tr.all())))
    assert flags == [(False, False), (False, False), (False, True)] * 2 + [(False, False)]
    env.close()


MAX_DISCRETE = {
    # S = 255 (the largest state id a history byte holds), 15 independent sets of 17 states
    "s255": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=255,
                 action_space_size=17, diameter=15, sequence_length=2, delay=3, reward_density=0.02, seed=1),
    # L = 7: 8^7 sequence keys, the reward bitmask (256 KiB) stays in HBM; delay 32 = longest shift register
    "l7": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
               action_space_size=8, sequence_length=7, delay=32, reward_density=0.02,
               terminal_state_density=0.125, seed=2),
    # delay 40 > 32: sequence keys wait in the HBM ring instead of the shift register; P-noise on
    "d40": dict(state_space_type="discrete", action_space_type="discrete", state_space_size=16,
                action_space_size=16, sequence_length=2, delay=40, transition_noise=0.1,
                terminal_state_density=0.0625, reward_density=0.5, seed=3),
}


@pytest.mark.parametrize("variant", sorted(MAX_DISCRETE))
def test_discrete_maximum_sizes_vs_oracle(variant):
    cfg = MAX_DISCRETE[variant]
    N, T = 320, 150
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = _venv(num_envs=N, autoreset="same_step", **cfg)
    A = cfg["action_space_size"]
    acts = np.random.default_rng(4).integers(0, A, size=(T, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
    obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    paid = 0
    for i in range(0, N, 9):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert o.reset() == int(init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        exp = eo.copy()
        exp[ed] = ero[ed]
        assert np.array_equal(obs[:, i], exp) and np.array_equal(term[:, i], ed), (variant, i)
        assert np.array_equal(rew[:, i], er.astype(np.float32)), (variant, i)
        paid += int((er != 0).sum())
    assert paid > 0 or variant == "l7"          # (a 7-state sequence is rarely completed by random actions)
    env.close()


def test_continuous_maximum_sizes_vs_oracle():
    """D = 32 (MDPP_MAX_DIM) with order 2, and order 4 (MDPP_MAX_ORDER) with 8 terminal hypercubes
    (MDPP_MAX_BOXES), delay 5, both noises."""
    big = dict(state_space_type="continuous", state_space_dim=32, relevant_indices=list(range(0, 32, 4)),
               irrelevant_features=True, target_point=[0.5] * 8, target_radius=0.4, state_space_max=6,
               action_space_max=1, transition_dynamics_order=2, inertia=1.5, time_unit=0.5, make_denser=True,
               reward_function="move_to_a_point", reward_noise=0.1, delay=5, seed=4)
    boxes = dict(state_space_type="continuous", state_space_dim=3, target_point=[0.0, 0.0, 0.0],
                 target_radius=0.3, state_space_max=4, action_space_max=1, transition_dynamics_order=4,
                 inertia=1.0, time_unit=0.7, make_denser=False, reward_function="move_to_a_point",
                 transition_noise=0.02, terminal_states=[[x, y, z] for x in (-2.5, 2.5) for y in (-2.5, 2.5)
                                                         for z in (-2.5, 2.5)],
                 term_state_edge=2.0, term_state_reward=-2.0, reward_scale=0.5, seed=5)
    for cfg in (big, boxes):
        N, T, D = 192, 80, cfg["state_space_dim"]
        env = _venv(num_envs=N, autoreset="same_step", **cfg)
        acts = np.random.default_rng(9).uniform(-1, 1, size=(T, N, D)).astype(np.float32)
        init = env._obs.cpu().numpy().copy()
        obs, rew, term, trunc = env.rollout(torch.as_tensor(acts, device=env.device))
        obs, rew, term = obs.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        for i in range(0, N, 11):
            o = _oracle_for(env, i)
            o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            assert np.array_equal(o.reset(), init[i])
            eo, er, ed, ero = o.rollout(acts[:, i], None)
            exp = eo.copy()
            exp[ed] = ero[ed]
            assert np.array_equal(obs[:, i].view(np.uint32), exp.view(np.uint32)), (D, i)
            assert np.array_equal(term[:, i], ed), (D, i)
            assert np.array_equal(rew[:, i], er.astype(np.float32)), (D, i)
        env.close()


def test_device_normals_match_numpy_stream():
    """The device ziggurat (chord/tangent pre-test + exp fallback) makes numpy's decisions: the
    reward-noise stream of 4096 envs x 400 steps (1.6 M normals incl. wedge and tail cases) is
    compared with numpy's Generator.normal on the same PCG64 states."""
    cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8,
               action_space_size=8, delay=0, sequence_length=1, reward_noise=1.0,
               terminal_state_density=0.0, reward_density=0.0, seed=5)
    N, T = 4096, 400
    env = _venv(num_envs=N, autoreset="disabled", **cfg)
    acts = torch.zeros((T, N), dtype=torch.int32, device=env.device)
    _, rew, _, _ = env.rollout(acts)
    rew = rew.cpu().numpy()
    from mdp_playground_amd import mdp as mdp_mod
    for i in range(0, N, 97):
        g = mdp_mod.new_generator(5 + i)
        g.random()                       # the construction-time reset() draw
        z = g.normal(0, 1.0, T)
        # the (single) rewardable state may add 1.0 before the noise: reward = float32(bit + z)
        ok = (rew[:, i] == z.astype(np.float32)) | (rew[:, i] == (1.0 + z).astype(np.float32))
        assert ok.all(), (i, int((~ok).sum()))
    env.close()


CFAST_GEN = dict(delay=3, reward_every_n_steps=2, reward_scale=1.5, reward_shift=-0.25, term_state_reward=-1.0,
                 target_radius=0.5, terminal_states=[[6.0, 6.0, 6.0, 6.0], [-5.0, 5.0, -5.0, 5.0]], term_state_edge=6.0)


@pytest.mark.parametrize("name", ["c_cfg5", "c_cfg3", "c_cfg5+gen", "c_cfg3+gen", "c_cfg3+unbounded"])
def test_continuous_fast_kernel_shared_vs_oracle(name):
    """BASELINE cfg 3 / cfg 5 shapes at 512 envs (full 256-env blocks, 40 fused steps): with noise
    this runs the producer/consumer variant (helper waves draw the normals); every sampled env
    must equal its oracle bit for bit, including the env stream's end state.  "+gen": the same
    with a reward delay, every-n 2, an affine map and two terminal hypercubes (the kernel's general
    post-processing: the reward's float32 / Python-float typing of the reference, resets that
    resample out of the cubes), across two launches so that the delay ring's head carries over."""
    gen = name.endswith("+gen")
    cfg = _cfg(name.split("+")[0], 23)
    if gen:
        cfg.update(CFAST_GEN)
    if name.endswith("+unbounded"):                 # no state_space_max: reset() samples normals (rolled loop via LDS)
        cfg.pop("state_space_max")
        cfg["target_radius"] = 1.5
    N, T = 512, 40
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=11, **cfg)
    rng = np.random.default_rng(6)
    acts = rng.uniform(-1, 1, size=(T, N, 12)).astype(np.float32)
    acts[7, 5, 3] = 2.0                      # one rejected action ("stay")
    init = env._obs.cpu().numpy().copy()
    assert env.rollout_kernel_name(T).startswith("k_continuous_rollout_fast<")
    at = torch.as_tensor(acts, device=env.device)
    parts = [env.rollout(at[:17]), env.rollout(at[17:])]          # (17 steps: not a multiple of the delay)
    obs, rew, term, trunc = (torch.cat([p[j] for p in parts]).cpu().numpy() for j in range(4))
    end_env = env.get_rng_streams(0)
    for i in list(range(0, N, 29)) + [5]:
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(T):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            tr = n >= 11
            if ed or tr:
                eo = o.reset()
                n = 0
            assert np.array_equal(obs[t, i], eo), (name, i, t)
            assert rew[t, i] == np.float32(er) and bool(trunc[t, i]) == tr, (name, i, t)
        assert np.array_equal(o.get_rng()[0][:4], end_env[i][:4]), (name, i)
    assert env.status()[5] == 1
    env.close()


@pytest.mark.parametrize("D,nrel,order", [(4, 2, 1), (4, 2, 2), (8, 4, 1), (8, 4, 2), (8, 8, 2), (2, 2, 2)])
def test_continuous_fast_kernel_other_shapes_vs_oracle(D, nrel, order):
    """The other (D, relevant dims, order) instantiations of k_continuous_rollout_fast, with both noises,
    a delay and truncation resets, on 256 envs against the oracle."""
    cfg = dict(state_space_type="continuous", state_space_dim=D, relevant_indices=list(range(nrel)),
               irrelevant_features=nrel < D, target_point=[0.5] * nrel, target_radius=0.3, state_space_max=6,
               action_space_max=1, transition_dynamics_order=order, inertia=2.0, time_unit=0.5, make_denser=True,
               reward_function="move_to_a_point", transition_noise=0.03, reward_noise=0.2, delay=2, seed=12)
    N, T = 256, 48
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=13, **cfg)
    assert env.rollout_kernel_name(T).startswith("k_continuous_rollout_fast<")
    acts = np.random.default_rng(3).uniform(-1, 1, size=(T, N, D)).astype(np.float32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
    for i in range(0, N, 37):
        o = _oracle_for(env, i)
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        assert np.array_equal(o.reset(), init[i])
        n = 0
        for t in range(T):
            eo, er, _, ed = o.step(acts[t, i])
            n += 1
            tr = n >= 13
            if ed or tr:
                eo = o.reset()
                n = 0
            assert np.array_equal(obs[t, i], eo), (i, t)
            assert rew[t, i] == np.float32(er) and bool(trunc[t, i]) == tr and bool(term[t, i]) == ed, (i, t)
    env.close()


LEAN_SHAPES = {
    # S <= 8, every L / delay / truncation branch of k_discrete_rollout_lean, one non-power-of-two S, int32 obs
    "l3_d4": (dict(state_space_size=8, action_space_size=8, delay=4, sequence_length=3), None, "int64"),
    "l1_d0": (dict(state_space_size=8, action_space_size=8, delay=0, sequence_length=1, terminal_state_density=0.25), None, "int64"),
    "l2_d1_max": (dict(state_space_size=8, action_space_size=5, delay=1, sequence_length=2, terminal_state_density=0.25,
                       reward_every_n_steps=1), 13, "int64"),
    "l2_d3_every5": (dict(state_space_size=8, action_space_size=8, delay=3, sequence_length=2, reward_every_n_steps=5), None, "int64"),
    "s5_l3_d2_max": (dict(state_space_size=5, action_space_size=3, delay=2, sequence_length=3), 7, "int32"),
    "s6_l2_d0_max": (dict(state_space_size=6, action_space_size=6, delay=0, sequence_length=2, terminal_state_density=0.34), 40, "int64"),
    "s3_l1_d32": (dict(state_space_size=3, action_space_size=2, delay=32, sequence_length=1), 9, "int32"),
    # an irrelevant sub-space (action / observation pairs, start states drawn in pairs)
    "irr_l3_d4": (dict(state_space_size=[8, 8], action_space_size=[8, 5], irrelevant_features=True, delay=4, sequence_length=3), None, "int64"),
    "irr_s5_l2_max": (dict(state_space_size=[5, 7], action_space_size=[3, 4], irrelevant_features=True, delay=1, sequence_length=2,
                           terminal_state_density=0.4), 11, "int32"),
}


@pytest.mark.parametrize("rng_mode", ["numpy", "philox"])
@pytest.mark.parametrize("shape", sorted(LEAN_SHAPES))
def test_lean_rollout_kernel_vs_oracle_and_other_kernels(shape, rng_mode):
    """k_discrete_rollout_lean (nibble history, v_perm transitions, one gated reward-bit table; see
    mdpp_discrete_lean.hip) on 1024 envs, several launches with a ragged last chunk: every output and the
    stream end states equal the pipelined and the single-role kernels of the old encoding, a strided
    sample equals the oracle step for step, and single steps continue from the state it leaves."""
    extra, max_steps, odt = LEAN_SHAPES[shape]
    cfg = dict(state_space_type="discrete", action_space_type="discrete", seed=23, **extra)
    N, launches = (1000 if shape in ("l1_d0", "s5_l3_d2_max", "l2_d3_every5") else 1024), 3      # (1000: a ragged last block)
    Ks = [40, 32, 77]
    kw = dict(num_envs=N, autoreset="same_step", rng=rng_mode, **cfg)
    if max_steps:
        kw["max_episode_steps"] = max_steps
    if odt == "int32":
        kw["dtype_o"] = np.int32
    envs = [_venv(**kw) for _ in range(3)]
    assert envs[0].rollout_kernel_name(64).startswith("k_discrete_rollout_lean<"), envs[0].rollout_kernel_name(64)
    if rng_mode == "numpy" and "irr" in shape:          # an irrelevant sub-space: against the quiet and the general kernel
        assert "IRR=1" in envs[0].rollout_kernel_name(64)
        envs[1].set_kernel_options("NO_LEAN")
        envs[2].set_kernel_options("NO_LEAN", "NO_QUIET")
        assert envs[1].rollout_kernel_name(64).startswith("k_discrete_rollout_quiet<")
        assert envs[2].rollout_kernel_name(64).startswith("k_discrete_step<")
    elif rng_mode == "numpy":
        envs[1].set_kernel_options("NO_LEAN")
        envs[2].set_kernel_options("NO_PIPE", "NO_HELPER")
        assert envs[1].rollout_kernel_name(64).startswith("k_discrete_rollout_pipe<" if N % 256 == 0 else "k_discrete_rollout_fast<")
        assert envs[2].rollout_kernel_name(64).startswith("k_discrete_rollout_fast<")
    else:       # Philox streams: the H waves make every tick's start state; against the quiet and the general kernel
        assert "PHILOX=1" in envs[0].rollout_kernel_name(64)
        envs[1].set_kernel_options("NO_LEAN")
        envs[2].set_kernel_options("NO_PHILOX_FAST")
        assert envs[1].rollout_kernel_name(64).startswith("k_discrete_rollout_quiet<")
        assert envs[2].rollout_kernel_name(64).startswith("k_discrete_step<")
    irr = isinstance(cfg["action_space_size"], list)
    A, A1 = (cfg["action_space_size"] if irr else (cfg["action_space_size"], None))
    rng = np.random.default_rng(5)
    init = envs[0]._obs.cpu().numpy().copy()
    outs = []
    for j in range(launches):
        acts = rng.integers(0, A, size=(Ks[j], N)).astype(np.int32)
        acts[3, 5] = -1                                   # negative index wraps (numpy semantics)
        if j == 1 and rng_mode == "philox":
            acts[9, 70] = A + 3                           # out of range: flagged, stepped as action 0 by every kernel
            acts[11, N - 1] = -A - 1                      # (the last env of the ragged block)
        if irr:
            acts1 = rng.integers(0, A1, size=(Ks[j], N)).astype(np.int32)
            acts1[4, 6] = -2
            acts = np.stack([acts, acts1], axis=2)
        ta = torch.as_tensor(acts, device=envs[0].device)
        res = [tuple(x.cpu().numpy() for x in e.rollout(ta)) for e in envs]
        for r in res[1:]:
            for x, y in zip(res[0], r):
                assert np.array_equal(x, y), (shape, j)
        outs.append((acts, res[0]))
    if rng_mode == "numpy":
        st = [e.get_rng_streams(0) for e in envs]
        assert np.array_equal(st[0], st[1]) and np.array_equal(st[0], st[2])
    # single steps from the state the lean kernel left == single steps from the state the old kernels left
    one = rng.integers(0, A, size=(N,)).astype(np.int32)
    if irr:
        one = np.stack([one, rng.integers(0, A1, size=(N,)).astype(np.int32)], axis=1)
    acts = torch.as_tensor(one, device=envs[0].device)
    for _ in range(6):
        r = [tuple(x.cpu().numpy() for x in e.step(acts)[:4]) for e in envs]
        for q in r[1:]:
            for x, y in zip(r[0], q):
                assert np.array_equal(x, y), shape
    for i in (range(0, N, 97) if rng_mode == "numpy" else []):      # (Philox: the general kernel above is held to the oracle elsewhere)
        o = _oracle_for(envs[0], i)
        o.set_rng(envs[0].seeded_streams[0][i], envs[0].seeded_streams[1][i])
        assert np.array_equal(np.asarray(o.reset()), init[i])
        n = 0
        for acts_j, (obs, rew, term, trunc) in outs:
            for t in range(acts_j.shape[0]):
                if irr:
                    a_t = [int(x) for x in acts_j[t, i]]
                    a_t = [a_t[0] if a_t[0] >= 0 else a_t[0] + A, a_t[1] if a_t[1] >= 0 else a_t[1] + A1]
                    eo, er, ed = o.step(np.asarray(a_t, dtype=np.int32))
                else:
                    a_t = int(acts_j[t, i])
                    eo, er, ed = o.step(a_t if a_t >= 0 else a_t + A)
                n += 1
                tr = bool(max_steps) and n >= max_steps
                if ed or tr:
                    eo = o.reset(explicit=False) if irr else o.reset()
                    n = 0
                assert np.array_equal(np.asarray(obs[t, i]), np.asarray(eo)), (shape, i, t)
                assert rew[t, i] == np.float32(er) and bool(trunc[t, i]) == tr and bool(term[t, i]) == bool(ed), (shape, i, t)
    for e in envs:
        st = e.status()
        if rng_mode == "philox":
            assert st[70] == 1 and st[N - 1] == 1 and (np.delete(st, [70, N - 1]) == 0).all()   # MDPP_STATUS_BAD_ACTION, those envs only
        else:
            assert (st == 0).all()
        e.close()


@pytest.mark.parametrize("rng_mode", ["numpy", "philox"])
@pytest.mark.parametrize("shape", ["l3_d4", "l1_d0", "irr_l3_d4"])
def test_lean_rollout_kernel_without_autoreset_vs_single_role_kernel(shape, rng_mode):
    """autoreset="disabled" (what the reference itself does: it keeps stepping from the terminal state until the caller
    resets): the lean kernel with its H lanes idle against the kernel the golden parity tests run on, masked resets
    by the caller in between, stream end states."""
    extra, _, _ = LEAN_SHAPES[shape]
    cfg = dict(state_space_type="discrete", action_space_type="discrete", seed=31, **extra)
    N = 2048
    a = _venv(num_envs=N, autoreset="disabled", rng=rng_mode, **cfg)
    b = _venv(num_envs=N, autoreset="disabled", rng=rng_mode, **cfg)
    b.set_kernel_options("NO_LEAN", "NO_PIPE", "NO_QUIET")
    assert a.rollout_kernel_name(64).startswith("k_discrete_rollout_lean<"), a.rollout_kernel_name(64)
    assert not b.rollout_kernel_name(64).startswith("k_discrete_rollout_lean<")
    irr = isinstance(cfg["action_space_size"], list)
    A, A1 = (cfg["action_space_size"] if irr else (cfg["action_space_size"], None))
    g = torch.Generator(device=a.device)
    g.manual_seed(3)
    for K in (40, 33, 64):
        acts = torch.randint(0, A, (K, N), generator=g, device=a.device, dtype=torch.int32)
        if irr:
            acts = torch.stack([acts, torch.randint(0, A1, (K, N), generator=g, device=a.device, dtype=torch.int32)], dim=2)
        ra, rb = a.rollout(acts), b.rollout(acts)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (shape, K)
        assert bool(ra[2][-1].any())                              # terminal envs keep reporting terminated
        mask = ra[2][-1].clone()
        oa, ob = a.reset(mask=mask)[0], b.reset(mask=mask)[0]
        assert torch.equal(oa, ob)
    if rng_mode == "numpy":
        assert np.array_equal(a.get_rng_streams(0), b.get_rng_streams(0))
    assert (a.status() == 0).all() and (b.status() == 0).all()
    a.close(); b.close()


@pytest.mark.parametrize("rng_mode", ["numpy", "philox"])
@pytest.mark.parametrize("shape", sorted(LEAN_SHAPES))
def test_lean_rollout_kernel_next_step_autoreset_vs_general_kernel(shape, rng_mode):
    """autoreset="next_step" on the lean kernel (the pending flag in bit 31 of the step counter, the reset call's
    zero reward through a zeroed history word) against k_discrete_step, which the oracle-loop test of
    test_gpu_boundary.py holds to the oracle: fused launches, single steps in between (the flag crosses kernels
    both ways), an out-of-range action on some steps (an error only where the call is not a reset), stream ends."""
    extra, max_steps, odt = LEAN_SHAPES[shape]
    cfg = dict(state_space_type="discrete", action_space_type="discrete", seed=29, **extra)
    N = 1000
    kw = dict(num_envs=N, autoreset="next_step", rng=rng_mode, **cfg)
    if max_steps:
        kw["max_episode_steps"] = max_steps
    if odt == "int32":
        kw["dtype_o"] = np.int32
    a, b = _venv(**kw), _venv(**kw)
    b.set_kernel_options("NO_LEAN", "NO_QUIET")
    assert a.rollout_kernel_name(64).startswith("k_discrete_rollout_lean<"), a.rollout_kernel_name(64)
    assert b.rollout_kernel_name(64).startswith("k_discrete_step<"), b.rollout_kernel_name(64)
    irr = isinstance(cfg["action_space_size"], list)
    A, A1 = (cfg["action_space_size"] if irr else (cfg["action_space_size"], None))
    rng = np.random.default_rng(9)

    def actions(K):
        x = rng.integers(0, A, size=(K, N)).astype(np.int32)
        x[rng.integers(0, K, size=40), rng.integers(0, N, size=40)] = A + 1      # out of range here and there
        if irr:
            x = np.stack([x, rng.integers(0, A1, size=(K, N)).astype(np.int32)], axis=2)
        return torch.as_tensor(x, device=a.device)
    ended = 0
    for K in (45, 1, 1, 64, 1, 33):
        acts = actions(K)
        if K == 1:
            ra, rb = a.step(acts[0])[:4], b.step(acts[0])[:4]
        else:
            ra, rb = a.rollout(acts), b.rollout(acts)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (shape, K)
        ended += int((ra[2] | ra[3]).sum())
    assert ended > 0
    sa, sb = a.status(), b.status()                 # (reading clears)
    assert np.array_equal(sa, sb) and (sa != 0).any()
    if rng_mode == "numpy":
        assert np.array_equal(a.get_rng_streams(0), b.get_rng_streams(0))
    a.close(); b.close()


SOAK_IRR = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=[8, 8],
                action_space_size=[8, 8], irrelevant_features=True, delay=4, sequence_length=3)


@pytest.mark.parametrize("name,flag", [("d_cfg2", "NO_PIPE"), ("d_cfg2", "NO_LEAN"), ("c_cfg5", "NO_HELPER"), ("c_cfg5", "NO_PARK"),
                                       ("irr", "NO_LEAN"), ("irr", "NO_LEAN,NO_DUO"), ("irr+pn", "NO_TRIO"), ("irr+pn+rn", "NO_DUO")])
def test_multi_wave_kernels_equal_single_role_kernels_soak(name, flag):
    """The producer/consumer kernels (LDS rings between waves) against the single-role kernels of
    the same arithmetic, full size, many launches: any lost or duplicated hand-off would show.
    (c_cfg5 / NO_PARK: the drifting producer lanes against the lockstep producer; irr*: the two-
    and three-role forms of the quiet discrete kernel against the smaller ones.)  The single-role side
    is selected per handle with mdpp_set_options (include/mdpp.h MDPP_OPT_*)."""
    if name.startswith("irr"):
        cfg = dict(SOAK_IRR, seed=31)
        if "+pn" in name:
            cfg["transition_noise"] = 0.15
        if "+rn" in name:
            cfg["reward_noise"] = 0.25
    else:
        cfg = _cfg(name, 31)
    N, F, launches = 65536, 256, 12
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    b.set_kernel_options(*flag.split(","))          # (irr: the lean kernel against the three-role / single-role quiet kernel)
    assert a.rollout_kernel_name(F) != b.rollout_kernel_name(F) or flag == "NO_PARK", (a.rollout_kernel_name(F), flag)
    g = torch.Generator(device=a.device)
    g.manual_seed(1)
    for j in range(launches):
        if name.startswith("irr"):
            acts = torch.randint(0, 8, (F, N, 2), generator=g, device=a.device, dtype=torch.int32)
        elif a.kind == "discrete":
            acts = torch.randint(0, 8, (F, N), generator=g, device=a.device, dtype=torch.int32)
        else:
            acts = torch.rand((F, N, 12), generator=g, device=a.device) * 2 - 1
        ra = a.rollout(acts)
        rb = b.rollout(acts)
        torch.cuda.synchronize()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (name, j)
    assert np.array_equal(a.get_rng_streams(0), b.get_rng_streams(0))
    assert (a.status() == 0).all() and (b.status() == 0).all()
    a.close(); b.close()


@pytest.mark.parametrize("workload,over,flag,N,F", [
    # whole-row stores of the lean kernel (one wave writes the block's piece of a row; the flag bytes of four envs packed with
    # v_perm_b32, rewards staged through LDS): the TimeLimit byte, int32 observations, F with a ragged last chunk, next-step mode
    ("cfg2", {"max_episode_steps": 13}, "NO_LEAN", 65536, 132),
    ("cfg2", {"max_episode_steps": 13, "dtype_o": np.int32, "delay": 0}, "NO_LEAN,NO_PIPE", 32768, 128),
    ("cfg2", {"autoreset": "next_step", "max_episode_steps": 9}, "NO_LEAN", 32768, 132),
    ("cfg2", {"rng": "philox", "max_episode_steps": 13, "reward_every_n_steps": 1}, "NO_LEAN", 32768, 132),
    ("cfg2_noise", {}, "NO_LEAN,NO_QUIET", 65536, 128),              # numpy streams: lean kernel with noise (H: both streams, by position) vs general
    ("cfg2_noise", {}, "NO_LEAN", 65536, 128),                       # ... vs the noisy quiet kernel (two roles)
    ("cfg2_noise", {"reward_noise": None, "max_episode_steps": 13}, "NO_LEAN", 32768, 128),          # transition noise only (start-state queue + noise bytes)
    ("cfg2_noise", {"transition_noise": None, "reward_every_n_steps": 3, "delay": 0}, "NO_LEAN", 32768, 128),   # reward noise only
    ("cfg2_noise", {"autoreset": "disabled", "terminal_state_density": 0.0}, "NO_LEAN,NO_QUIET", 32768, 128),   # (no resets: H still makes the noise)
    ("cfg2_noise", {"transition_noise": 0.5}, "NO_QUIET", 32768, 128),   # (rows' thresholds differ: the lean kernel declines, quiet vs general)
    ("cfg2_irr", {"transition_noise": 0.1}, "NO_QUIET", 65536, 128),
    # round 5: one MDP per env with each lane's tables in its own LDS slot (k_discrete_step<...,LDSTAB=2>) vs the same kernel
    # gathering them from L2 (NO_QUIET switches the slots off): unit and reward_dist rewards, reward noise, a ragged batch
    ("cfg2_per_env", {}, "NO_QUIET", 2048, 96),
    ("cfg2_per_env", {"reward_dist": [0.1, 1], "sequence_length": 1, "reward_noise": 0.2, "delay": 2, "max_episode_steps": 9}, "NO_QUIET", 1000, 64),
    # round 5: rewards that are not all 1.0 (reward_dist) on the role-split quiet kernel (float64 table in LDS, key delay line in
    # HBM) vs the general kernel -- the shapes of the reference's rainbow_reward_dist / dqn_delay_50_states sweeps
    ("d_s24_rdist", {}, "NO_QUIET", 65536, 128),
    ("d_s24_rdist", {"delay": 3, "sequence_length": 2, "reward_scale": 2.5, "reward_shift": -0.75, "term_state_reward": -1.5,
                     "max_episode_steps": 11}, "NO_QUIET", 32768, 132),
    ("d_s24_rdist", {"delay": 2, "transition_noise": 0.1, "reward_noise": 0.2, "reward_every_n_steps": 2}, "NO_QUIET", 32768, 128),
    ("d_s50_delay4", {"reward_dist": [0.25, 1], "sequence_length": 2, "reward_density": 0.01, "autoreset": "disabled"}, "NO_QUIET", 16384, 64),
    # ... and rho_0 through the bucket table (S > 16, round 5): unit rewards, 24 / 50 / 200 states, against the general kernel's search
    ("d_s50_delay4", {}, "NO_QUIET", 65536, 128),
    ("d_s50_delay4", {"state_space_size": 120, "action_space_size": 120, "terminal_state_density": 0.6, "delay": 1, "reward_noise": 0.1},
     "NO_QUIET", 16384, 128),
    ("d_s24_rdist", {"reward_dist": None, "terminal_state_density": 0.5}, "NO_QUIET", 32768, 128),
    ("grid", {}, "NO_GFAST", 65536, 128),                            # int64 pairs: one 128-bit store per step
    ("grid", {"irrelevant_features": True, "transition_noise": 0.2, "reward_noise": 0.1}, "NO_GFAST", 32768, 128),
    ("cfg3", {}, "NO_CFAST", 65536, 64),                             # transposed 128-bit stores
    ("cfg3", {"delay": 2, "reward_every_n_steps": 2, "state_space_dim": 4, "relevant_indices": [0, 1, 2, 3]},
     "NO_CFAST", 65536, 64),
    ("cfg5", {"delay": 3}, "NO_CFAST", 32768, 48),
    ("cfg4", {}, "NO_IMGFAST", 2048, 40),                            # fast vs general renderer, pipelined batches
    ("img100_all", {}, "NO_IMGFAST", 2048, 40),                      # ... the wide-template renderer (radii 10 ... 40: k_image_obs_wide)
    ("img100_all", {"image_transforms": "scale", "image_width": 96, "image_height": 120, "rng": "philox"}, "NO_IMGFAST", 1000, 20),
    # Philox streams: the fused kernels (normals made by producer waves / at the top of the step) vs the general ones
    ("cfg5", {"rng": "philox"}, "NO_PHILOX_FAST", 65536, 64),
    ("cfg5", {"rng": "philox"}, "NO_HELPER", 65536, 64),
    ("cfg5", {"rng": "philox", "delay": 3, "terminal_states": [[5.0, 5.0, 5.0, 5.0]], "term_state_edge": 6.0},
     "NO_PHILOX_FAST", 32768, 48),
    ("cfg3", {"rng": "philox"}, "NO_PHILOX_FAST", 65536, 64),
    ("cfg2_noise", {"rng": "philox"}, "NO_PHILOX_FAST", 65536, 128),   # lean kernel with noise (H: noise nibbles, O1: normals) vs general
    ("cfg2_noise", {"rng": "philox"}, "NO_LEAN", 65536, 128),          # ... vs the quiet kernel's producer waves
    ("cfg2_noise", {"rng": "philox", "reward_noise": None, "max_episode_steps": 13}, "NO_LEAN", 32768, 128),     # transition noise only
    ("cfg2_noise", {"rng": "philox", "transition_noise": None, "reward_every_n_steps": 3, "delay": 0}, "NO_LEAN", 32768, 128),  # reward noise only
    ("cfg2_noise", {"rng": "philox", "autoreset": "disabled", "terminal_state_density": 0.0}, "NO_LEAN", 32768, 128),    # (H makes the noise without resets)
    ("cfg2", {"rng": "philox"}, "NO_PHILOX_FAST", 65536, 128),     # lean kernel, H waves on Philox blocks, vs general
    ("cfg2", {"rng": "philox"}, "NO_LEAN", 65536, 128),            # ... vs the quiet kernel's producer waves
    ("cfg2_irr", {"rng": "philox", "transition_noise": 0.1, "reward_noise": 0.2}, "NO_PHILOX_FAST", 32768, 64),
    ("grid", {"rng": "philox"}, "NO_PHILOX_FAST", 65536, 128),
    ("grid", {"rng": "philox", "irrelevant_features": True, "transition_noise": 0.2, "reward_noise": 0.1}, "NO_PHILOX_FAST", 32768, 128),
    # next-step autoreset on the fused kernels (pending flag in the flags word; a reset call draws nothing from the noise streams)
    ("cfg2_noise", {"autoreset": "next_step", "max_episode_steps": 9}, "NO_QUIET", 16384, 128),       # (two roles)
    ("cfg2_noise", {"rng": "philox", "autoreset": "next_step"}, "NO_PHILOX_FAST", 16384, 128),         # (Philox producers)
    ("cfg2_irr", {"autoreset": "next_step", "state_space_size": [8, 11], "action_space_size": [8, 11]}, "NO_QUIET", 16384, 128),  # (three roles)
    ("cfg3", {"autoreset": "next_step", "max_episode_steps": 7}, "NO_CFAST", 16384, 64),
    ("cfg3", {"autoreset": "next_step", "max_episode_steps": 5, "delay": 2, "target_radius": 6.0}, "NO_CFAST", 16384, 64),
    ("cfg5", {"rng": "philox", "autoreset": "next_step", "max_episode_steps": 5}, "NO_PHILOX_FAST", 16384, 64),
    ("grid", {"autoreset": "next_step"}, "NO_GFAST", 16384, 128),
    ("grid", {"autoreset": "next_step", "transition_noise": 0.2, "reward_noise": 0.1, "max_episode_steps": 11}, "NO_GFAST", 16384, 128),
    ("grid", {"rng": "philox", "autoreset": "next_step", "transition_noise": 0.2}, "NO_PHILOX_FAST", 16384, 128),
])
def test_specialised_kernels_equal_general_kernels_all_envs(workload, over, flag, N, F):
    """Every specialised rollout kernel against the general kernel of the same arithmetic, on EVERY env
    of a large batch (the oracle tests sample envs): outputs, and the streams' end states.  A
    lane-pattern fault such as the 128-bit store-data hazard shows up here."""
    import bench
    from mdp_playground_amd import _capi as capi
    wl = bench.WORKLOADS[workload]
    over = dict(over)
    rng = over.pop("rng", "numpy")
    ekw = dict(autoreset=over.pop("autoreset", "same_step"), max_episode_steps=over.pop("max_episode_steps", None))
    cfg = dict(wl["config"], **over)
    cfg = {k: v for k, v in cfg.items() if v is not None}       # (None: drop the key from the workload's config)
    nkw = dict(seeds=list(range(N))) if wl.get("per_env_mdps") else dict(num_envs=N)
    a = _venv(rng=rng, **nkw, **ekw, **cfg)
    b = _venv(rng=rng, **nkw, **ekw, **cfg)
    b.set_kernel_options(*flag.split(","))
    assert a.rollout_kernel_name(F) != b.rollout_kernel_name(F), (a.rollout_kernel_name(F), flag)
    wl2 = dict(wl, config=cfg)
    ended = 0
    for j in range(3):
        acts = bench.make_actions(wl2, F, N, a.device, 100 + j)
        ra = a.rollout(acts)
        rb = b.rollout(acts)
        torch.cuda.synchronize()
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (workload, j)
        ended += int((ra[2] | ra[3]).sum())
    if ekw["autoreset"] == "next_step":
        assert ended > 0                                         # (some episodes ended: the reset calls were exercised)
    streams = [capi.STREAM_ENV, capi.STREAM_SPACE] if rng == "numpy" else []       # (Philox: no stream state)
    if a.kind == "grid":
        streams.append(capi.STREAM_ACTION)
    if a.kind == "discrete" and a._irr:
        streams.append(capi.STREAM_SPACE_IRR)
    if getattr(a, "_image", None) is not None and a.kind == "discrete":
        streams.append(capi.STREAM_IMAGE)
    for st in (streams if rng == "numpy" else []):
        assert np.array_equal(a.get_rng_streams(st), b.get_rng_streams(st)), (workload, st)
    assert (a.status() == 0).all() and (b.status() == 0).all()
    a.close(); b.close()


@pytest.mark.parametrize("what", ["states", "images"])
def test_rollout_is_graph_capturable(what):
    """include/mdpp.h promises that nothing is allocated inside mdpp_step / mdpp_step_n: a fused
    rollout can be captured into a HIP graph and replayed.  (The handle's step counter travels by
    value, so a replayed graph is exact for numpy-stream handles with unit rewards — the counter only
    feeds Philox keys and the key ring of non-unit rewards.)  "images": the two-stream batch pipeline
    of image rollouts forks to the handle's side stream and joins back inside the capture."""
    cfg = _cfg("d_cfg2", 23) if what == "states" else dict(IMG_CFGS["cfg4"], seed=23)
    N, K = (4096, 64) if what == "states" else (512, 40)
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    acts = torch.randint(0, 8, (K, N), device=a.device, dtype=torch.int32)
    out_a = a.alloc_rollout(K)
    side = torch.cuda.Stream(device=a.device)
    side.wait_stream(torch.cuda.current_stream(a.device))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        a.rollout(acts, out_a)                     # warm-up outside the capture
    torch.cuda.current_stream(a.device).wait_stream(side)
    torch.cuda.synchronize()
    b.rollout(acts)                                # keep b in step with a's warm-up
    with torch.cuda.graph(g, stream=side):
        a.rollout(acts, out_a)
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        ob, rb, tb, _ = b.rollout(acts)
        assert torch.equal(out_a[0], ob) and torch.equal(out_a[1], rb) and torch.equal(out_a[2].view(torch.bool), tb), rep
    a.close(); b.close()


# ----------------------------------------------------------------------------- one launch = one step (round 5)
_S1_D_CFG2 = dict(gu.CASES["d_cfg2"]["config"], seed=3)
_S1_C_CFG3 = dict(gu.CASES["c_cfg3"]["config"], seed=3)
_S1_C_CFG5 = dict(gu.CASES["c_cfg5"]["config"], seed=3)
STEP1_CASES = {
    # name: (config, constructor keywords, N, expected kernel prefix or None)
    "d_cfg2_numpy": (_S1_D_CFG2, dict(autoreset="same_step"), 4096, "k_discrete_step1<OBS64=1,PHILOX=0>"),
    "d_cfg2_philox": (_S1_D_CFG2, dict(autoreset="same_step", rng="philox"), 4096, "k_discrete_step1<OBS64=1,PHILOX=1>"),
    "d_cfg2_philox_ragged_disabled": (_S1_D_CFG2, dict(autoreset="disabled", rng="philox"), 1000, "k_discrete_step1<"),
    "d_s16_trunc": (FAST_VARIANTS["s16_l1_trunc"][0], dict(autoreset="same_step", max_episode_steps=5), 1000, "k_discrete_step1<"),
    "d_s16_trunc_philox": (FAST_VARIANTS["s16_l1_trunc"][0], dict(autoreset="same_step", max_episode_steps=5, rng="philox"), 1000,
                           "k_discrete_step1<"),
    "d_s6_nonpow2_everyn": (dict(FAST_VARIANTS["s6_l2_nonpow2"][0], reward_every_n_steps=3), dict(autoreset="same_step"), 1000,
                            "k_discrete_step1<"),
    # state spaces beyond 16 states (k_discrete_step1w; the reference's 24- and 50-state sweeps)
    "d_s50_numpy": (dict(__import__("bench").WORKLOADS["d_s50_delay4"]["config"], seed=3), dict(autoreset="same_step"), 4096,
                    "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=1>"),
    "d_s50_philox_ragged_trunc": (dict(__import__("bench").WORKLOADS["d_s50_delay4"]["config"], seed=3, sequence_length=2, reward_density=0.02),
                                  dict(autoreset="same_step", rng="philox", max_episode_steps=6), 1000, "k_discrete_step1w<OBS64=1,PHILOX=1,UNIT=1>"),
    "d_s24_unit_disabled": (dict(__import__("bench").WORKLOADS["d_s24_rdist"]["config"], seed=3, reward_dist=None, delay=3, sequence_length=3,
                                 reward_every_n_steps=2), dict(autoreset="disabled"), 1000, "k_discrete_step1w<"),
    "d_s24_rdist": (dict(__import__("bench").WORKLOADS["d_s24_rdist"]["config"], seed=3), dict(autoreset="same_step"), 4096,
                    "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=0>"),
    "d_s24_rdist_delay_philox": (dict(__import__("bench").WORKLOADS["d_s24_rdist"]["config"], seed=3, delay=3, sequence_length=2, reward_scale=2.5,
                                      reward_shift=-0.75, term_state_reward=-1.5), dict(autoreset="same_step", rng="philox", max_episode_steps=7), 1000,
                                 "k_discrete_step1w<OBS64=1,PHILOX=1,UNIT=0>"),
    "d_s24_rdist_delay_disabled": (dict(__import__("bench").WORKLOADS["d_s24_rdist"]["config"], seed=3, delay=2, reward_every_n_steps=3),
                                   dict(autoreset="disabled"), 1000, "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=0>"),
    # ... with transition / reward noise (the reference's p_noise / r_noise sweeps): numpy streams and Philox streams
    "d_cfg2_noise_numpy": (dict(__import__("bench").WORKLOADS["cfg2_noise"]["config"], seed=3), dict(autoreset="same_step"), 4096,
                           "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=1,PN=1,RN=1>"),
    "d_cfg2_noise_philox_ragged": (dict(__import__("bench").WORKLOADS["cfg2_noise"]["config"], seed=3), dict(autoreset="same_step", rng="philox", max_episode_steps=9),
                                   1000, "k_discrete_step1w<OBS64=1,PHILOX=1,UNIT=1,PN=1,RN=1>"),
    "d_s16_pnoise_only": (dict(FAST_VARIANTS["s16_l1_trunc"][0], transition_noise=0.3), dict(autoreset="same_step", max_episode_steps=5), 1000,
                          "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=1,PN=1,RN=0>"),
    "d_rnoise_only_disabled": (dict(__import__("bench").WORKLOADS["cfg2_noise"]["config"], seed=3, transition_noise=None, reward_scale=2.5, reward_shift=-1.0,
                                    term_state_reward=-0.5), dict(autoreset="disabled"), 1000, "k_discrete_step1w<OBS64=1,PHILOX=0,UNIT=1,PN=0,RN=1>"),
    "d_s120": (dict(state_space_type="discrete", action_space_type="discrete", state_space_size=120, action_space_size=60, delay=1,
                    sequence_length=1, terminal_state_density=0.5, seed=3), dict(autoreset="same_step"), 1024, "k_discrete_step1w<"),
    # polygon pictures: draw + record + render in one kernel (k_image_step1) against the four launches
    "i_cfg4": (dict(IMG_CFGS["cfg4"], seed=3), dict(autoreset="same_step"), 1000, "k_image_step1<NST=7>"),
    "i_cfg4_bench_size": (dict(__import__("bench").WORKLOADS["cfg4"]["config"]), dict(autoreset="same_step"), 8192, "k_image_step1<NST=7>"),   # (1.6 rounds of waves on the chip)
    "i_img100_all_bench_size": (dict(__import__("bench").WORKLOADS["img100_all"]["config"]), dict(autoreset="same_step"), 8192, "k_image_step1<WIDE=1>"),
    "i_cfg4_philox_trunc": (dict(IMG_CFGS["cfg4"], seed=3), dict(autoreset="same_step", rng="philox", max_episode_steps=5), 1000, "k_image_step1<NST=7>"),
    "i_cfg4_disabled": (dict(IMG_CFGS["cfg4"], seed=3), dict(autoreset="disabled"), 333, "k_image_step1<NST=7>"),
    "i_cfg4_next_step": (dict(IMG_CFGS["cfg4"], seed=3), dict(autoreset="next_step", max_episode_steps=6), 333, "k_image_step1<NST=7>"),
    "i_rot64": (dict(IMG_CFGS["rot64"], seed=3), dict(autoreset="same_step"), 1000, "k_image_step1<NST=4>"),
    "i_irr84": (dict(IMG_CFGS["irr84"], seed=3), dict(autoreset="same_step"), 500, "k_image_step1<NST=7>"),      # two pictures per env, one generator
    "i_irr84_philox_trunc": (dict(IMG_CFGS["irr84"], seed=3), dict(autoreset="same_step", rng="philox", max_episode_steps=4), 333, "k_image_step1<NST=7>"),
    "i_img100_all_wide": (dict(__import__("bench").WORKLOADS["img100_all"]["config"], seed=3), dict(autoreset="same_step"), 500, "k_image_step1<WIDE=1>"),
    "i_scale_only_wide_philox": (dict(__import__("bench").WORKLOADS["img100_all"]["config"], seed=3, image_transforms="scale", image_width=96, image_height=120),
                                 dict(autoreset="same_step", rng="philox", max_episode_steps=5), 333, "k_image_step1<WIDE=1>"),
    "i_scale_flip_96x80": (dict(IMG_CFGS["cfg4"], seed=3, image_width=96, image_height=80, image_transforms="shift,scale,rotate,flip",
                                image_scale_range=(0.5, 1.0), image_sh_quant=2, image_ro_quant=15), dict(autoreset="same_step"), 500, "k_image_step1<NST=0>"),
    "c_cfg3": (_S1_C_CFG3, dict(autoreset="same_step"), 4096, "k_continuous_step1<D=12,ORDER=1,NREL=4,NOISE=0,GEN=0,PHILOX=0,WG=64>"),
    "c_cfg3_ragged_next_step": (_S1_C_CFG3, dict(autoreset="next_step", max_episode_steps=7), 1000, "k_continuous_step1<"),
    "c_cfg3_disabled": (_S1_C_CFG3, dict(autoreset="disabled"), 1000, "k_continuous_step1<"),
    "c_cfg5_numpy": (_S1_C_CFG5, dict(autoreset="same_step"), 4096, "k_continuous_step1<D=12,ORDER=2,NREL=4,NOISE=1,GEN=0,PHILOX=0,PAR=1>"),
    "c_cfg5_numpy_ragged": (_S1_C_CFG5, dict(autoreset="same_step"), 1000, "k_continuous_rollout_fast<"),   # (numpy noise: whole 64-env groups only)
    "c_cfg5_numpy_sequential": (_S1_C_CFG5, dict(autoreset="same_step"), 1024, "k_continuous_step1<D=12,ORDER=2,NREL=4,NOISE=1,GEN=0,PHILOX=0,WG=256>"),
    "c_cfg5_numpy_heavy_noise": (dict(_S1_C_CFG5, transition_noise=3.0, reward_noise=1.0, state_space_max=4), dict(autoreset="same_step"), 4096,
                                 "k_continuous_step1<D=12,ORDER=2,NREL=4,NOISE=1,GEN=0,PHILOX=0,PAR=1>"),
    "c_pnoise_only_d2": (dict(gu.CASES["c_default_target_sparse"]["config"], target_point=[0.5, -0.5], seed=3), dict(autoreset="same_step"), 1024,
                         "k_continuous_step1<D=2,ORDER=1,NREL=2,NOISE=1,GEN=1,PHILOX=0,WG=256>"),      # (PAR is for D >= 8: round 5)
    "c_pnoise_only_d8": (dict(state_space_type="continuous", state_space_dim=8, transition_dynamics_order=1, inertia=1, time_unit=0.5, state_space_max=5,
                              action_space_max=1, target_point=[0.5] * 8, target_radius=1.0, make_denser=True, reward_function="move_to_a_point",
                              transition_noise=0.3, seed=3), dict(autoreset="same_step"), 1024,
                         "k_continuous_step1<D=8,ORDER=1,NREL=8,NOISE=1,GEN=0,PHILOX=0,PAR=1>"),
    "c_cfg5_philox": (_S1_C_CFG5, dict(autoreset="same_step", rng="philox"), 1000, "k_continuous_step1<"),
    "c_boxes": (dict(gu.CASES["c_sparse_term"]["config"], seed=3), dict(autoreset="same_step"), 1000, "k_continuous_step1<D=2,ORDER=1,NREL=2,NOISE=0,GEN=1"),
    "c_everyn_rnoise": (dict(gu.CASES["c_small_radius_hit"]["config"], seed=3), dict(autoreset="same_step"), 1024, "k_continuous_step1<D=2,ORDER=1,NREL=2,NOISE=1,GEN=1"),
    "c_delay_order2": (dict(state_space_type="continuous", state_space_dim=4, transition_dynamics_order=2, inertia=2.0, time_unit=0.5,
                            delay=2, action_space_max=1, state_space_max=3, target_point=[0.5, -0.5, 0.0, 1.0], target_radius=0.7,
                            make_denser=True, reward_function="move_to_a_point", action_loss_weight=0.05, reward_scale=1.5,
                            reward_shift=-0.25, term_state_reward=2.0, transition_noise=0.05, seed=3),
                       dict(autoreset="same_step", rng="philox"), 1000, "k_continuous_step1<D=4,ORDER=2,NREL=4,NOISE=1,GEN=1,PHILOX=1"),
}


@pytest.mark.parametrize("case", sorted(STEP1_CASES))
def test_step1_kernels_equal_the_rollout_kernels_with_k1(case):
    """mdpp_step's own kernels (k_discrete_step1, k_continuous_step1: every load of the launch issued at once, no
    rollout prologue) against the kernels that served single steps until round 4 (the rollout kernels with K = 1, selected
    with NO_STEP1; those are pinned to the oracle and the goldens by the tests above): every output of every step, the end
    state and every stream's end state, bit for bit -- single steps, a fused piece in between (shared start-state queue),
    rejected actions (continuous: "stay" makes the kernel read the rows it otherwise skips; discrete: the status bit)."""
    cfg, kw, N, prefix = STEP1_CASES[case]
    cfg = {k: v for k, v in cfg.items() if v is not None}
    a, b = _venv(num_envs=N, **kw, **cfg), _venv(num_envs=N, **kw, **cfg)
    b.set_kernel_options("NO_STEP1")
    if case == "c_cfg5_numpy_sequential":       # the one-step kernel that draws its normals one after the other in the env's lane
        a.set_kernel_options("NO_HELPER")
    assert a.rollout_kernel_name(1).startswith(prefix), a.rollout_kernel_name(1)
    assert "step1" not in b.rollout_kernel_name(1)
    T = 48
    g = np.random.default_rng(5)
    if a.kind == "discrete":
        A = a.mdps[0].A
        acts = g.integers(0, A, size=(3 * T, N)).astype(np.int32)
        acts[g.random((3 * T, N)) < 0.01] = -1          # numpy's negative index: the last action
        acts[g.random((3 * T, N)) < 0.002] = A + 3      # out of range: action 0 and MDPP_STATUS_BAD_ACTION
        if a._irr:                                      # (Tuple spaces: one action per sub-space)
            acts = np.stack([g.integers(0, A, size=(3 * T, N)), g.integers(0, a.mdps[0].A_irr, size=(3 * T, N))], axis=2).astype(np.int32)
    else:
        D = a.mdps[0].D
        acts = g.uniform(-1, 1, size=(3 * T, N, D)).astype(np.float32)
        bad = g.random((3 * T, N)) < 0.01
        acts[bad, g.integers(0, D, size=int(bad.sum()))] = 1.5      # outside the action box: the env stays where it is
    acts = torch.as_tensor(acts, device=a.device)

    def same(x, y):
        return torch.equal(x.view(torch.int32) if x.dtype.is_floating_point else x, y.view(torch.int32) if y.dtype.is_floating_point else y)
    for t in list(range(T)) + list(range(2 * T, 3 * T)):
        if t == 2 * T:
            ra, rb = a.rollout(acts[T:2 * T]), b.rollout(acts[T:2 * T])
            assert all(same(x, y) for x, y in zip(ra, rb)), (case, "fused piece")
        ra, rb = a.step(acts[t]), b.step(acts[t])
        assert all(same(x, y) for x, y in zip(ra[:4], rb[:4])), (case, t)
        if case.startswith("i_"):                       # (the terminal pictures of the envs this step has reset; other rows stay as they were)
            assert ("final_obs" in ra[4]) == ("final_obs" in rb[4]) == (kw["autoreset"] == "same_step")
            assert "final_obs" not in ra[4] or torch.equal(ra[4]["final_obs"], rb[4]["final_obs"]), (case, t, "final_obs")
    assert np.array_equal(a.status(), b.status())
    sa, sb = a.get_augmented_state(), b.get_augmented_state()
    for k in sa:
        if isinstance(sa[k], np.ndarray):
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (case, k)
    if kw.get("rng", "numpy") == "numpy":
        for s in (0, 1) + ((2,) if case.startswith("i_") else ()):
            assert np.array_equal(a.get_rng_streams(s), b.get_rng_streams(s)), (case, s)
    a.close(); b.close()
