"""GPU tests of the drop-in boundary itself: autoreset modes, seed(), batched
spaces, the per-handle kernel switches and kernel names, state import validation, the replayable
graph of single steps.  Everything goes through the C ABI (RLToyVectorEnv -> libmdpp_hip.so)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from test_gpu_parity import _oracle_for, _venv

pytestmark = pytest.mark.gpu


def _set_oracle_streams(env, o, i):
    from mdp_playground_amd import _capi as capi
    if env.kind == "grid":
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i], env.seeded_streams[capi.STREAM_ACTION][i])
    else:
        o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
        if env.kind == "discrete" and env._irr:
            o.set_rng_irr(env.seeded_streams[capi.STREAM_SPACE_IRR][i])


def _rand_actions(env, T, rng):
    N = env.num_envs
    if env.kind == "discrete" and env._irr:
        m = env.mdps[0]
        return np.stack([rng.integers(0, m.A, size=(T, N)), rng.integers(0, m.A_irr, size=(T, N))], axis=2).astype(np.int32)
    if env.kind == "discrete":
        return rng.integers(0, env.mdps[0].A, size=(T, N)).astype(np.int32)
    if env.kind == "grid":
        G = len(env.mdps[0].grid_shape)
        ac = np.zeros((T, N, G), np.int32)
        np.put_along_axis(ac, rng.integers(0, G, size=(T, N, 1)), rng.integers(-1, 2, size=(T, N, 1)).astype(np.int32), axis=2)
        return ac
    return rng.uniform(-1, 1, size=(T, N, env.mdps[0].D)).astype(np.float32)


def _oracle_step(o, kind, a):
    if kind == "continuous":
        ob, r, _, d = o.step(a)
        return ob, r, d
    return o.step(a if kind != "discrete" or np.ndim(a) else int(a))


@pytest.mark.parametrize("name,max_steps", [("d_cfg2", 0), ("d_cfg2_noise", 7), ("d_irr_noise", 0), ("c_cfg5", 5),
                                            ("c_sparse_term", 0), ("g_noise_sparse", 0)])
@pytest.mark.parametrize("fused", [False, True])
def test_next_step_autoreset_vs_oracle_loop(name, max_steps, fused):
    """autoreset="next_step" (gymnasium >= 1.0 vector envs): the call after an episode's last step
    ignores that env's action, calls reset() and returns (first obs, 0.0, False, False) — against the
    oracle driven by exactly that loop; noisy configs, so the streams must not move on the reset call."""
    cfg = dict(gu.CASES[name]["config"], seed=13)
    N, T = 384, 60
    env = _venv(num_envs=N, autoreset="next_step", max_episode_steps=max_steps or None, **cfg)
    fused_kernel = {"d_cfg2": "k_discrete_rollout_lean<",          # at most 8 states, no noise
                    "c_sparse_term": "k_continuous_rollout_fast<",  # no noise (numpy noise streams are drawn ahead per step)
                    "g_noise_sparse": "k_grid_rollout_fast<",
                    "d_cfg2_noise": "k_discrete_rollout_quiet<",    # inline draws, skipped on a reset call
                    "d_irr_noise": "k_discrete_rollout_quiet<"}.get(name)
    if fused_kernel:
        assert env.rollout_kernel_name(T).startswith(fused_kernel), env.rollout_kernel_name(T)
    else:                                                          # general kernels serve the mode
        assert "rollout" not in env.rollout_kernel_name(T)
    rng = np.random.default_rng(3)
    acts = _rand_actions(env, T, rng)
    init = env._obs.cpu().numpy().copy()
    acts_t = torch.as_tensor(acts, device=env.device)
    if fused:
        obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(acts_t))
    else:
        outs = []
        for t in range(T):
            o_, r_, te_, tr_, info = env.step(acts_t[t])
            assert "final_obs" not in info
            outs.append((o_.cpu().numpy().copy(), r_.cpu().numpy().copy(), te_.cpu().numpy().copy(), tr_.cpu().numpy().copy()))
        obs, rew, term, trunc = (np.stack(x) for x in zip(*outs))
    assert (term | trunc).any()
    for i in range(0, N, 5):
        o = _oracle_for(env, i)
        _set_oracle_streams(env, o, i)
        assert np.array_equal(np.asarray(o.reset()), init[i])
        pending, n = False, 0
        for t in range(T):
            if pending:
                eo, er, ed, etr = o.reset(), 0.0, False, False
                pending, n = False, 0
            else:
                eo, er, ed = _oracle_step(o, env.kind, acts[t, i])
                n += 1
                etr = bool(max_steps) and n >= max_steps
                pending = ed or etr
            assert np.array_equal(np.asarray(eo), obs[t, i]), (name, i, t)
            assert rew[t, i] == np.float32(er) and bool(term[t, i]) == ed and bool(trunc[t, i]) == etr, (name, i, t)
    assert (env.status() == 0).all()
    env.close()


def test_next_step_state_roundtrip_clears_nothing_it_should_not():
    """get/set_augmented_state on a next_step handle: the pending flag is not part of the exported
    counters (steps stay small), and an explicit reset() clears it."""
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=2)
    env = _venv(num_envs=256, autoreset="next_step", **cfg)
    acts = torch.randint(0, 8, (40, 256), device=env.device, dtype=torch.int32)
    _, _, term, _ = env.rollout(acts)
    st = env.get_augmented_state()
    assert st["total_transitions_episode"].max() < 64
    env.reset()
    o, r, te, tr, _ = env.step(acts[0])
    # after an explicit reset nobody is pending: a pending env would return reward 0 and its reset obs
    # with steps 0; here every env made one transition
    assert (env.get_augmented_state()["total_transitions_episode"] == 1).all()
    env.close()


@pytest.mark.parametrize("name", ["d_cfg2", "d_cfg2_noise", "c_sparse_term", "g_noise_sparse"])
def test_next_step_checkpoint_between_the_last_step_and_the_reset_call(name):
    """a checkpoint taken right after a terminal step (next-step autoreset: the NEXT call is the reset)
    carries the per-env pending flag through get/set_augmented_state: a rollout split by a checkpoint into a fresh
    env equals the unsplit one, at every split point of a stretch in which episodes end."""
    from mdp_playground_amd import _capi as capi
    cfg = dict(gu.CASES[name]["config"], seed=21)
    N, T = 256, 24
    mk = lambda: _venv(num_envs=N, autoreset="next_step", max_episode_steps=5, **cfg)   # noqa: E731
    ref = mk()
    acts = torch.as_tensor(_rand_actions(ref, T, np.random.default_rng(8)), device=ref.device)
    want = [x.clone() for x in ref.rollout(acts)]
    streams = [capi.STREAM_ENV, capi.STREAM_SPACE] + ([capi.STREAM_ACTION] if ref.kind == "grid" else [])
    seen_pending = False
    for split in (5, 6, 11, 17):
        a, b = mk(), mk()
        head = a.rollout(acts[:split])
        st = a.get_augmented_state()
        assert st["reset_pending"].dtype == bool and st["reset_pending"].shape == (N,)
        seen_pending = seen_pending or bool(st["reset_pending"].any())
        b.set_augmented_state(st)
        for s_ in streams:
            b._put_stream(s_, a.get_rng_streams(s_))
        tail = b.rollout(acts[split:])
        for w, h, t in zip(want, head, tail):
            assert torch.equal(w[:split], h) and torch.equal(w[split:], t), (name, split)
        # without the flag the restored env keeps stepping from the terminal state: the export must matter
        if st["reset_pending"].any():
            c = mk()
            c.set_augmented_state({k: v for k, v in st.items() if k != "reset_pending"})
            assert not c.get_augmented_state()["reset_pending"].any()
            c.close()
        a.close(); b.close()
    assert seen_pending
    ref.close()


def test_reset_pending_refused_without_next_step_autoreset():
    from mdp_playground_amd import _capi as capi
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=2)
    env = _venv(num_envs=64, autoreset="same_step", **cfg)
    st = env.get_augmented_state()
    assert "reset_pending" not in st
    env.set_augmented_state(st)
    with pytest.raises(capi.MdppError):
        env.set_augmented_state(dict(st, reset_pending=np.ones(64, bool)))
    env.close()


def test_seed_returns_seed_and_reseeds_env_streams():
    """seed(s) -> s (rl_toy_env.py:2379-2406); env i continues from PCG64(SeedSequence(s + i))."""
    from mdp_playground_amd import mdp as mdp_mod
    cfg = dict(gu.CASES["d_cfg2_noise"]["config"], seed=4)
    N, T = 128, 30
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    assert env.seed(4242) == 4242
    with pytest.raises(TypeError):
        env.seed(-1)
    assert isinstance(env.seed(), int)
    assert env.seed(99) == 99
    space_before = env.get_rng_streams(1)
    ob, _ = env.reset()
    ob = ob.cpu().numpy()
    acts = np.random.default_rng(0).integers(0, 8, size=(T, N)).astype(np.int32)
    obs, rew, term, _ = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
    for i in range(0, N, 9):
        o = _oracle_for(env, i)
        o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator(99 + i)), space_before[i])
        assert o.reset() == ob[i]
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(obs[:, i], eo) and np.array_equal(rew[:, i], er.astype(np.float32)), i
    env.close()


def test_batched_spaces():
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=1)
    env = _venv(num_envs=16, **cfg)
    assert env.observation_space.shape == (16,) and env.action_space.shape == (16,)
    a = env.action_space.sample()
    assert a.shape == (16,) and env.action_space.contains(a) and not env.action_space.contains(a + 100)
    env.step(a)
    assert env.single_action_space.n == 8
    env.close()
    env = _venv(num_envs=4, **dict(gu.CASES["c_cfg3"]["config"], seed=1))
    assert env.observation_space.shape == (4, 12) and env.action_space.sample().shape == (4, 12)
    assert env.observation_space.contains(env._obs.cpu().numpy())
    env.close()


def test_kernel_names_and_options():
    """mdpp_kernel_name reports what the library's own dispatch launches; mdpp_set_options takes
    specialised kernels out per handle (no process-global switches)."""
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=1)
    a = _venv(num_envs=65536, autoreset="same_step", **cfg)
    b = _venv(num_envs=65536, autoreset="same_step", **cfg)
    assert a.rollout_kernel_name(512) == "k_discrete_rollout_lean<OBS64=1,DELAY=1,HASMAX=0,EVN=1,PHILOX=0,IRR=0,NEXT=0>"
    b.set_kernel_options("NO_LEAN")
    assert b.rollout_kernel_name(512) == "k_discrete_rollout_pipe<OBS64=1,POW2=1,DELAY=1,S8=1>"
    assert a.rollout_kernel_name(16).startswith("k_discrete_rollout_fast<") and "HELPER=0" in a.rollout_kernel_name(16)
    b.set_kernel_options("NO_PIPE")
    assert b.rollout_kernel_name(512).startswith("k_discrete_rollout_fast<") and "HELPER=1" in b.rollout_kernel_name(512)
    assert a.rollout_kernel_name(512).startswith("k_discrete_rollout_lean<")       # a is unaffected
    b.set_kernel_options("NO_PIPE", "NO_HELPER")
    assert "HELPER=0" in b.rollout_kernel_name(512)
    b.set_kernel_options()
    assert b.rollout_kernel_name(512) == a.rollout_kernel_name(512)
    a.close(); b.close()
    c = _venv(num_envs=1000, autoreset="same_step", **cfg)                         # ragged batch: lean takes it, pipe does not
    assert c.rollout_kernel_name(512).startswith("k_discrete_rollout_lean<")
    c.set_kernel_options("NO_LEAN")
    assert c.rollout_kernel_name(512).startswith("k_discrete_rollout_fast<")
    d = _venv(num_envs=200, autoreset="same_step", **cfg)                          # less than one block: single-role kernel
    assert d.rollout_kernel_name(512).startswith("k_discrete_rollout_fast<")
    d.close()
    c.close()
    d = _venv(num_envs=512, rng="philox", **cfg)
    assert "k_discrete" in d.rollout_kernel_name(64)
    d.close()


@pytest.mark.parametrize("which", ["l4", "irr_l4"])
def test_reseed_with_partial_mask_keeps_long_histories(which):
    """reset(seed=s, mask=partial) re-seeds every env but resets only the masked ones: the unmasked
    envs' history bytes 4-7 (sequence_length >= 4; general / quiet kernels) must survive the re-seed
    (the queue word of the packed-nibble kernels is cleared only for those handles)."""
    from mdp_playground_amd import mdp as mdp_mod
    if which == "l4":
        cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
                   sequence_length=5, delay=2, reward_density=0.5, terminal_state_density=0.125, seed=3)
    else:
        cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=[8, 6],
                   action_space_size=[8, 6], irrelevant_features=True, sequence_length=4, delay=1,
                   reward_density=0.5, terminal_state_density=0.125, seed=3)
    N, T = 256, 48
    env = _venv(num_envs=N, autoreset="disabled", **cfg)
    assert not env.rollout_kernel_name(T).startswith(("k_discrete_rollout_fast", "k_discrete_rollout_pipe"))
    rng = np.random.default_rng(1)
    acts1, acts2 = _rand_actions(env, 3, rng), _rand_actions(env, T, rng)     # 3 steps: histories still hold NaN slots
    init = env._obs.cpu().numpy().copy()
    seeded = {k: v.copy() for k, v in env.seeded_streams.items()}       # (reset(seed=) below replaces the record)
    obs1, *_ = env.rollout(torch.as_tensor(acts1, device=env.device))
    mask = rng.random(N) < 0.5
    ob, _ = env.reset(seed=555, mask=torch.as_tensor(mask, device=env.device))
    ob = ob.cpu().numpy()
    obs2, rew2, term2, _ = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts2, device=env.device)))
    env.seeded_streams = seeded
    for i in range(0, N, 3):
        o = _oracle_for(env, i)
        _set_oracle_streams(env, o, i)
        assert np.array_equal(np.asarray(o.reset()), init[i])
        for t in range(3):
            o.step(acts1[t, i])
        w_sp = o.get_rng()[1]
        o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator(555 + i)), w_sp)      # seed: env stream only
        if mask[i]:
            assert np.array_equal(np.asarray(o.reset()), ob[i]), i
        for t in range(T):
            eo, er, ed = o.step(acts2[t, i])
            assert np.array_equal(np.asarray(eo), obs2[t, i]), (i, t)
            assert rew2[t, i] == np.float32(er) and bool(term2[t, i]) == ed, (i, t)
    assert (env.status() == 0).all()
    env.close()


def test_set_state_validation_and_ring_import_of_non_unit_rewards():
    from mdp_playground_amd import _capi as capi
    cfg = dict(gu.CASES["d_rdist"]["config"], seed=5)                 # non-unit rewards (reward_dist)
    cfg["delay"] = 3
    N = 256
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    acts = torch.as_tensor(np.random.default_rng(2).integers(0, a.mdps[0].A, size=(40, N)).astype(np.int32), device=a.device)
    a.rollout(acts[:20])
    st = a.get_augmented_state()
    assert (st["reward_buffer"] != 0).any()                           # something is waiting in the delay line
    assert len(np.unique(st["reward_buffer"])) > 2                    # ... and not only unit rewards
    b.set_augmented_state(st)                                         # values -> keys that pay exactly those values
    for s_ in (0, 1):
        b._put_stream(s_, a.get_rng_streams(s_))
    assert np.array_equal(b.get_augmented_state()["reward_buffer"], st["reward_buffer"])
    ra, rb = a.rollout(acts[20:]), b.rollout(acts[20:])
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    bad = dict(st, reward_buffer=st["reward_buffer"] + 0.123456)       # a value no sequence pays
    with pytest.raises(capi.MdppError):
        b.set_augmented_state(bad)
    hist = st["augmented_state"].copy()
    hist[0, 0], hist[0, 1] = 2, -1                                    # a NaN slot newer than a valid state
    with pytest.raises(capi.MdppError):
        b.set_augmented_state(dict(st, augmented_state=hist))
    hist = st["augmented_state"].copy()
    hist[3, :] = -1                                                   # no current state
    with pytest.raises(capi.MdppError):
        b.set_augmented_state(dict(st, augmented_state=hist))
    a.close(); b.close()


def test_step_graph_replays_single_steps():
    """step_graph(K): K captured mdpp_step launches == K step() calls; new actions written into the
    captured tensor are picked up by the next replay."""
    cfg = dict(gu.CASES["d_cfg2"]["config"], seed=9)
    N, K = 4096, 32
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    acts = torch.randint(0, 8, (K, N), device=a.device, dtype=torch.int32)
    g = a.step_graph(acts)
    for rep in range(3):
        if rep:
            g.actions.copy_(torch.randint(0, 8, (K, N), device=a.device, dtype=torch.int32))
        g.replay()
        torch.cuda.synchronize()
        for t in range(K):
            o, r, te, tr, _ = b.step(g.actions[t])
            assert torch.equal(o, g.obs[t]) and torch.equal(r, g.reward[t]) and torch.equal(te, g.terminated[t]), (rep, t)
    a.close(); b.close()


def test_step_graph_counts_its_steps_and_refuses_inexact_handles():
    """the step counter travels by value into captured launches.  (1) The capture does not advance it and a
    replay advances it by K: plain steps after replays use the right delay-line slot (continuous env, delay 2, K = 4).
    (2) Philox handles and delay lines in memory whose length does not divide K are refused.  (3) A replay after a
    number of plain steps that moves the ring head is refused."""
    from mdp_playground_amd import _capi as capi
    cfg = dict(gu.CASES["c_order3_delay3"]["config"], seed=5)
    d = int(cfg["delay"])
    N, K = 512, 2 * d
    a = _venv(num_envs=N, autoreset="same_step", **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **cfg)
    rng = np.random.default_rng(3)
    acts = torch.as_tensor(_rand_actions(a, K, rng), device=a.device)
    g = a.step_graph(acts)
    for rep in range(3):
        if rep:
            g.actions.copy_(torch.as_tensor(_rand_actions(a, K, rng), device=a.device))
        g.replay()
        torch.cuda.synchronize()
        for t in range(K):
            o, r, te, tr, _ = b.step(g.actions[t])
            assert torch.equal(o, g.obs[t]) and torch.equal(r, g.reward[t]) and torch.equal(te, g.terminated[t]), (rep, t)
        # plain steps between replays: a whole number of ring turns keeps the graph usable
        for t in range(d):
            x = torch.as_tensor(_rand_actions(a, 1, rng)[0], device=a.device)
            oa, ra, _, _, _ = a.step(x)
            ob, rb, _, _, _ = b.step(x)
            assert torch.equal(oa, ob) and torch.equal(ra, rb), (rep, t)
    a.step(acts[0])                                         # one more: the ring head no longer matches the captured one
    with pytest.raises(capi.MdppError):
        g.replay()
    assert a._lib.mdpp_graph_replay_exact(a._h, d + 1) == 2   # K % delay != 0: exact through the device-side offset (below)
    a.close(); b.close()
    # image observations whose draws are keyed by the counter: refused until round 4, through the device-side offset now (below)
    icfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8, delay=0,
                image_representations=True, image_width=84, image_height=84, image_transforms="shift,rotate", seed=0)
    p = _venv(num_envs=256, autoreset="same_step", rng="philox", **icfg)
    assert p._lib.mdpp_graph_replay_exact(p._h, 4) == 2
    p.close()


@pytest.mark.parametrize("case", ["cfg2-philox", "cfg2noise-philox", "cfg3delay3-numpy", "cfg5-philox", "cfg5-numpy-delay2",
                                  "custom-reward-delay", "grid-philox", "s50-philox", "s24rdist-delay2-philox",
                                  "image-cfg4-philox", "image-cfg4-numpy", "image-irr84-philox", "image-continuous-philox",
                                  "image-grid-philox"])
def test_step_graph_exact_for_every_handle_through_the_tick_offset(case):
    """step() is the API RL code calls; a replayed graph of K single steps is exact for EVERY handle (round
    5: image observations too -- k_image_step1 / k_image_draw key a step's transforms by counter + device word).  Launches captured in the library's capture mode add a device word to the step counter they were
    captured with (Philox keys, the head of a delay line in memory); replay() sets it to (counter now - counter at
    capture).  Replays interleaved with plain steps -- any number of them, so the ring head and the Philox ticks move --
    equal a twin stepped call by call, bit for bit."""
    import bench
    if case == "cfg2-philox":
        cfg, kw = dict(bench.WORKLOADS["cfg2"]["config"]), dict(rng="philox", philox_seed=3)
    elif case == "cfg2noise-philox":
        cfg, kw = dict(bench.WORKLOADS["cfg2_noise"]["config"]), dict(rng="philox", philox_seed=4)
    elif case == "cfg3delay3-numpy":
        cfg, kw = dict(bench.WORKLOADS["cfg3"]["config"], delay=3), {}
    elif case == "cfg5-philox":
        cfg, kw = dict(bench.WORKLOADS["cfg5"]["config"]), dict(rng="philox", philox_seed=5)
    elif case == "cfg5-numpy-delay2":
        cfg, kw = dict(bench.WORKLOADS["cfg5"]["config"], delay=2), {}
    elif case == "s50-philox":                   # round 5: k_discrete_step1w, start states keyed by the tick
        cfg, kw = dict(bench.WORKLOADS["d_s50_delay4"]["config"]), dict(rng="philox", philox_seed=8)
    elif case == "s24rdist-delay2-philox":       # ... its float-reward form: the key delay line's head AND the Philox tick move
        cfg, kw = dict(bench.WORKLOADS["d_s24_rdist"]["config"], delay=2), dict(rng="philox", philox_seed=9)
    elif case == "custom-reward-delay":          # float rewards: the discrete delay line lives in memory (keys awaiting payout)
        cfg, kw = dict(gu.CASES["d_rdist"]["config"], seed=2, delay=3), {}
    elif case.startswith("image-cfg4"):          # polygon pictures: one kernel draws, records and renders (k_image_step1)
        cfg = dict(bench.WORKLOADS["cfg4"]["config"])
        kw = dict(rng="philox", philox_seed=10) if case.endswith("philox") else {}
    elif case == "image-irr84-philox":           # ... two pictures per env from one stream
        cfg, kw = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=[8, 11], action_space_size=[8, 11],
                       irrelevant_features=True, delay=0, image_representations=True, image_width=84, image_height=84,
                       image_transforms="shift,rotate", seed=2), dict(rng="philox", philox_seed=11)
    elif case == "image-continuous-philox":
        cfg = dict(state_space_type="continuous", state_space_dim=4, relevant_indices=[0, 1], transition_dynamics_order=2, inertia=1.0,
                   time_unit=1.0, state_space_max=4, action_space_max=1, make_denser=True, target_point=[1.5, -2.0], target_radius=0.7,
                   terminal_states=[[-2.0, 2.0], [3.0, 0.0]], term_state_edge=1.5, transition_noise=0.1, reward_noise=0.05,
                   reward_function="move_to_a_point", image_representations=True, image_width=64, image_height=80, seed=4)
        kw = dict(rng="philox", philox_seed=12)
    elif case == "image-grid-philox":
        cfg = dict(state_space_type="grid", grid_shape=(6, 5), reward_function="move_to_a_point", make_denser=True, target_point=[2, 3],
                   irrelevant_features=True, transition_noise=0.2, terminal_states=[[0, 0], [5, 4]], image_representations=True,
                   image_width=48, image_height=64, seed=8)
        kw = dict(rng="philox", philox_seed=13)
    else:
        cfg, kw = dict(bench.WORKLOADS["grid"]["config"], transition_noise=0.2, reward_noise=0.1), dict(rng="philox", philox_seed=6)
    N, K = (256 if case.startswith("image-") else 1024), 5
    a = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    b = _venv(num_envs=N, autoreset="same_step", **kw, **cfg)
    if case in ("custom-reward-delay", "s50-philox", "s24rdist-delay2-philox"):
        assert a.rollout_kernel_name(1).startswith("k_discrete_step1w<"), a.rollout_kernel_name(1)
    if case.startswith("image-cfg4"):
        assert a.rollout_kernel_name(1) == "k_image_step1<NST=7>"
    assert a._lib.mdpp_graph_replay_exact(a._h, K) == (1 if case == "image-cfg4-numpy" else 2), case
    rng = np.random.default_rng(7)
    g = a.step_graph(torch.as_tensor(_rand_actions(a, K, rng), device=a.device))
    for rep in range(4):
        g.actions.copy_(torch.as_tensor(_rand_actions(a, K, rng), device=a.device))
        g.replay()
        torch.cuda.synchronize()
        for t in range(K):
            o, r, te, tr, _ = b.step(g.actions[t])
            assert torch.equal(o, g.obs[t]) and torch.equal(r, g.reward[t]), (case, rep, t)
            assert torch.equal(te, g.terminated[t]) and torch.equal(tr, g.truncated[t]), (case, rep, t)
        for t in range(rep + 1):                 # plain steps in between: 1, 2, 3 ... (never a multiple of every delay)
            x = torch.as_tensor(_rand_actions(a, 1, rng)[0], device=a.device)
            ra, rb = a.step(x), b.step(x)
            assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]), (case, rep, t)
    assert int(a.status().sum()) == 0
    a.close(); b.close()


def test_large_action_space_tables_fall_back_to_global_memory():
    """S * A beyond the default dynamic-LDS limit: the step kernel reads the tables from HBM / L2
    instead of failing its first launch."""
    cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=250, action_space_size=250,
               sequence_length=1, delay=0, reward_density=0.1, terminal_state_density=0.05, seed=2)
    N, T = 128, 20
    env = _venv(num_envs=N, autoreset="same_step", **cfg)
    assert "LDSTAB=0" in env.rollout_kernel_name(1)
    acts = np.random.default_rng(4).integers(0, 250, size=(T, N)).astype(np.int32)
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, _ = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
    for i in range(0, N, 11):
        o = _oracle_for(env, i)
        _set_oracle_streams(env, o, i)
        assert o.reset() == int(init[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(obs[:, i], eo) and np.array_equal(rew[:, i], er.astype(np.float32)), i
    env.close()


def _stats_rows(st, kind, D=0):
    """get_episode_stats() dict -> [N, 4 (+ D)] in the order of the fixtures' `stats` rows."""
    cols = [st["total_abs_noise_in_reward_episode"], np.asarray(st["total_reward_episode"], dtype=np.float64),
            st["total_noisy_transitions_episode"].astype(np.float64) if kind != "continuous" else np.zeros_like(st["total_abs_noise_in_reward_episode"]),
            st["total_transitions_episode"].astype(np.float64)]
    out = np.stack(cols, axis=1)
    if kind == "continuous":
        out = np.concatenate([out, st["total_abs_noise_in_transition_episode"]], axis=1)
    return out


@pytest.mark.parametrize("name", ["d_stats", "g_stats", "c_stats"])
def test_episode_stats_vs_reference_golden(name):
    """episode_stats=True (VERDICT r2 "missing"): the reference's per-episode noise statistics per env instance, after every
    step of the reference's own trajectories -- float64 bit patterns of what the reference env object held
    (total_abs_noise_in_reward_episode, total_reward_episode, total_noisy_transitions_episode,
    total_abs_noise_in_transition_episode, total_transitions_episode) -- and after every reset() the figures it logged."""
    from test_gpu_parity import _seeds_or_cfg
    g = gu.load(name)
    E, T = g["action"].shape[:2]
    env = _venv(autoreset="disabled", episode_stats=True, **_seeds_or_cfg(name))
    assert "rollout" not in env.rollout_kernel_name(8)                     # the general kernels keep the statistics
    D = g["action"].shape[2] if env.kind == "continuous" else 0
    for t in range(T):
        env.step(torch.as_tensor(g["action"][:, t], device=env.device))
        st = env.get_episode_stats()
        got = _stats_rows(st, env.kind, D)
        assert np.array_equal(got.view(np.uint64), g["stats"][:, t].view(np.uint64)), (name, t, got[0], g["stats"][0, t])
        ra = g["reset_after"][:, t]
        if ra.any():
            env.reset(mask=torch.as_tensor(ra, device=env.device))
            st = env.get_episode_stats()
            last = _stats_rows(st["last_episode"], env.kind, D)
            assert np.array_equal(last[ra].view(np.uint64), g["stats"][:, t][ra].view(np.uint64)), (name, t)
            assert not _stats_rows(st, env.kind, D)[ra].any()
    env.close()


@pytest.mark.parametrize("name,rng", [("d_stats", "numpy"), ("d_stats", "philox"), ("g_stats", "numpy"), ("c_stats", "numpy"),
                                      ("c_stats", "philox")])
def test_episode_stats_fused_rollout_vs_oracle(name, rng):
    """The same statistics through a fused rollout with same-step autoreset (in-kernel resets roll them into
    `last_episode`), 1 024 envs of one MDP, every 9th env against the oracle: running episode and last finished episode."""
    cfg = dict(gu.CASES[name]["config"], seed=31)
    N, T = 1024, 70
    kw = dict(rng="philox", philox_seed=41) if rng == "philox" else {}
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=23, episode_stats=True, **kw, **cfg)
    acts = _rand_actions(env, T, np.random.default_rng(6))
    init = env._obs.cpu().numpy().copy()
    obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
    st = env.get_episode_stats()
    D = env.mdps[0].D if env.kind == "continuous" else 0
    cur, last = _stats_rows(st, env.kind, D), _stats_rows(st["last_episode"], env.kind, D)
    assert (term | trunc).any()
    for i in range(0, N, 9):
        o = _oracle_for(env, i)
        if rng == "philox":
            o.set_philox(41, i)
        else:
            _set_oracle_streams(env, o, i)
        assert np.array_equal(np.asarray(o.reset()), init[i])
        n = 0
        for t in range(T):
            _, er, ed = _oracle_step(o, env.kind, acts[t, i])
            n += 1
            assert np.float32(er) == rew[t, i] and bool(ed) == bool(term[t, i]), (name, i, t)
            if ed or n >= 23:
                o.reset(explicit=False)
                n = 0
        oc, ol = o.get_stats()
        want_cur = np.concatenate([oc[:2], [0.0 if env.kind == "continuous" else oc[2]], [n], oc[3:]])
        want_last = np.concatenate([ol[:2], [0.0 if env.kind == "continuous" else ol[2]], [ol[-1]], ol[3:-1]])
        assert np.array_equal(cur[i].view(np.uint64), want_cur.view(np.uint64)), (name, i, cur[i], want_cur)
        assert np.array_equal(last[i].view(np.uint64), want_last.view(np.uint64)), (name, i, last[i], want_last)
    env.close()
    with pytest.raises(Exception):
        _venv(num_envs=8, **cfg).get_episode_stats()


def test_episode_stats_with_image_observations():
    """episode_stats on handles with image observations (the state kernel of every batch keeps them): a polygon-image env and
    a grid-picture env hold the statistics of their integer-observation twins, a continuous-picture env the oracle's (with
    the reference's image-mode clipping, rl_toy_env.py:1601-1622), through fused rollouts with same-step autoreset."""
    T = 40
    dcfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
                delay=1, sequence_length=2, reward_density=0.25, terminal_state_density=0.25, transition_noise=0.2,
                reward_noise=0.5, seed=12)
    gcfg = dict(state_space_type="grid", grid_shape=(6, 5), reward_function="move_to_a_point", make_denser=True,
                target_point=[2, 3], transition_noise=0.2, terminal_states=[[0, 0], [5, 4]], seed=8)
    for cfg, img in ((dcfg, dict(image_representations=True, image_width=64, image_height=64, image_transforms="shift,rotate")),
                     (gcfg, dict(image_representations=True, image_width=48, image_height=64))):
        N = 512
        a = _venv(num_envs=N, autoreset="same_step", max_episode_steps=17, episode_stats=True, **cfg, **img)
        b = _venv(num_envs=N, autoreset="same_step", max_episode_steps=17, episode_stats=True, **cfg)
        acts = torch.as_tensor(_rand_actions(b, T, np.random.default_rng(2)), device=a.device)
        ra, rb = a.rollout(acts), b.rollout(acts)
        assert torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]) and ra[2].any()
        sa, sb = a.get_episode_stats(), b.get_episode_stats()
        for k in sb:
            if k == "last_episode":
                for k2 in sb[k]:
                    assert np.array_equal(np.asarray(sa[k][k2]).view(np.uint64), np.asarray(sb[k][k2]).view(np.uint64)), (k, k2)
            else:
                assert np.array_equal(np.asarray(sa[k]).view(np.uint64), np.asarray(sb[k]).view(np.uint64)), k
        assert np.asarray(sb["last_episode"]["total_transitions_episode"]).any()
        a.close(); b.close()
    ccfg = dict(state_space_type="continuous", state_space_dim=4, relevant_indices=[0, 1], transition_dynamics_order=2,
                inertia=1.0, time_unit=1.0, state_space_max=4, action_space_max=1, make_denser=True,
                target_point=[1.5, -2.0], target_radius=0.7, terminal_states=[[-2.0, 2.0], [3.0, 0.0]], term_state_edge=1.5,
                transition_noise=0.1, reward_noise=0.05, reward_function="move_to_a_point", image_representations=True,
                image_width=32, image_height=40, seed=4)
    N = 256
    env = _venv(num_envs=N, autoreset="same_step", max_episode_steps=17, episode_stats=True, **ccfg)
    acts = _rand_actions(env, T, np.random.default_rng(6))
    obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
    st = env.get_episode_stats()
    cur, last = _stats_rows(st, "continuous", 4), _stats_rows(st["last_episode"], "continuous", 4)
    assert term.any()
    for i in range(0, N, 5):
        o = _oracle_for(env, i)
        o.set_image_quirk(True)
        _set_oracle_streams(env, o, i)
        o.reset()
        n = 0
        for t in range(T):
            _, er, ed = _oracle_step(o, "continuous", acts[t, i])
            n += 1
            assert np.float32(er) == rew[t, i] and bool(ed) == bool(term[t, i]), (i, t)
            if ed or n >= 17:
                o.reset(explicit=False)
                n = 0
        oc, ol = o.get_stats()
        want_cur = np.concatenate([oc[:2], [0.0], [n], oc[3:]])
        want_last = np.concatenate([ol[:2], [0.0], [ol[-1]], ol[3:-1]])
        assert np.array_equal(cur[i].view(np.uint64), want_cur.view(np.uint64)), (i, cur[i], want_cur)
        assert np.array_equal(last[i].view(np.uint64), want_last.view(np.uint64)), (i, last[i], want_last)
    env.close()


@pytest.mark.parametrize("name", ["c_line_4d", "c_line_irr", "c_line_6of8"])
def test_line_reward_checkpoint_carries_the_window_of_the_fit(name):
    """move_along_a_line fits the last sequence_length states.  get_augmented_state() returns that window
    (the part of the reference's augmented_state list the reward reads, rl_toy_env.py:1865-1872, :2147-2156; NaN before
    the episode's reset), set_augmented_state() restores it into a fresh twin at every split point of a rollout, and
    the twin continues bit for bit like the uninterrupted handle -- the first L rewards after the restore included.
    Restoring the counters WITHOUT the window makes stepping fail loudly instead of fitting a stale one."""
    from mdp_playground_amd import RLToyVectorEnv, _capi
    import golden_util as gu
    cfg = dict(gu.CASES[name]["config"], seed=3)
    N, D = 192, cfg["state_space_dim"]
    L = cfg["sequence_length"]
    mk = lambda: RLToyVectorEnv(num_envs=N, autoreset="same_step", max_episode_steps=2 * L + 3, **cfg)   # noqa: E731
    env = mk()
    acts = torch.as_tensor(np.random.default_rng(5).uniform(-1, 1, size=(4 * L + 9, N, D)).astype(np.float32), device=env.device)
    for split in (1, L - 1, L, 2 * L + 4, 3 * L + 1):
        a, b = mk(), mk()
        for t in range(split):
            a.step(acts[t])
        st = a.get_augmented_state()
        n_rel = a._cfg.n_rel
        assert st["augmented_state"].shape == (N, L, n_rel)
        steps = st["total_transitions_episode"]
        live = np.arange(L)[None, :] + steps[:, None] + 1 >= L
        assert np.array_equal(~np.isnan(st["augmented_state"]).any(axis=2), live)           # NaN exactly before the episode
        rel = list(a.mdps[0].relevant_indices)
        assert np.array_equal(st["augmented_state"][:, -1], st["curr_state"][:, rel])       # newest = the current state
        b._lib.mdpp_tick(b._h, split, None)
        if split == L:
            # the REFERENCE's layout of augmented_state (a list of sequence_length + delay + 1 full state vectors per env,
            # rl_toy_env.py:660): accepted too -- the last L rows' relevant columns are the window
            d_ = cfg.get("delay", 0)
            full = np.full((N, L + d_ + 1, D), np.nan, np.float32)
            full[:, -L:, :][:, :, rel] = st["augmented_state"]
            b.set_augmented_state(dict(st, augmented_state=full))
        else:
            b.set_augmented_state(st)
        for s_ in (0, 1):
            b._put_stream(s_, a.get_rng_streams(s_))
        for t in range(split, split + L + 3):
            ra, rb = a.step(acts[t]), b.step(acts[t])
            for x, y in zip(ra[:4], rb[:4]):
                assert torch.equal(x, y), (name, split, t)
        # counters without the window: refused
        c = mk()
        rc = c._lib.mdpp_set_state_continuous(c._h, _capi.nptr(np.ascontiguousarray(st["state_derivatives"], np.float32)),
                                             _capi.nptr(np.ascontiguousarray(st["curr_state"], np.float32)),
                                             _capi.nptr(np.ascontiguousarray(steps, np.int32)), None, None, None)
        assert rc == 0
        with pytest.raises(_capi.MdppError, match="set_line_history"):
            c.step(acts[0])
        for e in (a, b, c):
            e.close()
    env.close()
