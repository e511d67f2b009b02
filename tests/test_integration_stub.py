"""The reference-side ctypes stub printed in INTEGRATION.md section 2 is executed, so that it cannot rot:
  * (CPU) its struct mirrors include/mdpp.h field for field (= the maintained binding's) and names the header's ABI;
  * (GPU) driven with a host env object of the reference's shape -- attributes of RLToyEnv filled from the host MDP
    generator for BASELINE cfg 2 -- it reproduces the oracle's trajectories and the maintained binding's, bit for bit."""
import os
import re
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class RLToyEnvHIP" in b]
    assert len(stub) == 1
    return stub[0]


def _load_stub():
    from mdp_playground_amd import _capi
    os.environ["MDPP_LIB"] = _capi.LIB_PATH
    mod = types.ModuleType("rl_toy_env_hip")
    exec(compile(_stub_source(), "INTEGRATION.md#stub", "exec"), mod.__dict__)
    return mod


def _header_abi():
    hdr = open(os.path.join(ROOT, "include", "mdpp.h")).read()
    return int(re.search(r"#define\s+MDPP_ABI_VERSION\s+(\d+)", hdr).group(1))


def test_stub_struct_mirrors_the_header_and_names_its_abi():
    from mdp_playground_amd import _capi
    mod = _load_stub()
    ours = [(n, t) for n, t in _capi.MdppConfig._fields_]
    theirs = [(n, t) for n, t in mod.MdppConfig._fields_]
    assert [n for n, _ in ours] == [n for n, _ in theirs]
    import ctypes as C
    for (n, a), (_, b) in zip(ours, theirs):
        assert C.sizeof(a) == C.sizeof(b), n
    assert C.sizeof(_capi.MdppConfig) == C.sizeof(mod.MdppConfig)
    for f in _capi.MdppConfig._fields_:
        assert getattr(_capi.MdppConfig, f[0]).offset == getattr(mod.MdppConfig, f[0]).offset, f[0]
    abi = _header_abi()
    assert abi == _capi.MDPP_ABI_VERSION
    src = _stub_source()
    assert f"abi_version={abi}," in src and f"mdpp_abi_version() == {abi}" in src


@pytest.mark.gpu
def test_stub_reproduces_cfg2_against_oracle_and_maintained_binding():
    import torch
    from mdp_playground_amd import RLToyVectorEnv, mdp
    from oracle import oracle as ora

    cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
               delay=4, sequence_length=3, seed=0)
    m = mdp.build_mdp(cfg)
    # a host object of the reference's shape (rl_toy_env.py attribute names), as RLToyEnv(**cfg) would be
    host = types.SimpleNamespace(
        state_space_size=[m.S], action_space_size=[m.A], sequence_length=m.sequence_length, delay=m.delay,
        reward_every_n_steps=m.reward_every_n_steps, reward_scale=m.reward_scale, reward_shift=m.reward_shift,
        term_state_reward=m.term_state_reward, transition_matrix=np.asarray(m.P), rewardable_sequences=m.rewardable_sequences,
        config={"terminal_states": list(m.terminal_states), "relevant_init_state_dist": np.asarray(m.init_dist)},
        seed_dict=dict(m.seed_dict))
    N, T = 96, 60
    mod = _load_stub()
    stub = mod.RLToyEnvHIP(host, N)
    ours = RLToyVectorEnv(num_envs=N, autoreset="same_step", **cfg)
    ours.reset(seed=m.seed_dict["env"])
    obs0 = stub.obs.cpu().numpy().copy()
    assert np.array_equal(obs0, ours._obs.cpu().numpy())
    acts = np.random.default_rng(11).integers(0, 8, size=(T, N)).astype(np.int32)
    got = []
    for t in range(T):
        a = torch.as_tensor(acts[t], device="cuda:0")
        o, r, tm, tr, _ = stub.step(a)
        o2, r2, tm2, tr2, _ = ours.step(a)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(tm, tm2)
        got.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), tm.cpu().numpy().copy()))
    for i in range(0, N, 7):
        o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P, m.reward_table(),
                               m.terminal_states, m.init_dist, m.transition_noise, m.reward_noise, m.reward_scale,
                               m.reward_shift, m.term_state_reward)
        w = mdp.pcg64_words(mdp.new_generator(m.seed_dict["env"] + i))
        o.set_rng(w, w)                       # (the space stream is unused without transition noise)
        assert o.reset() == int(obs0[i])
        eo, er, ed, ero = o.rollout(acts[:, i], None)
        eo[ed] = ero[ed]
        assert np.array_equal(np.array([g[0][i] for g in got]), eo)
        assert np.array_equal(np.array([g[1][i] for g in got]), er.astype(np.float32))
        assert np.array_equal(np.array([g[2][i] for g in got]).astype(bool), ed)
    stub.close()
    ours.close()
