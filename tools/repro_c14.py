#!/usr/bin/env python3
"""Repro (GPU box): continuous D = 14, order 3, Philox streams, TimeLimit 7 against the oracle, env by env."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np, torch
from mdp_playground_amd import _capi
if os.environ.get('MDPP_LIB'):
    _capi.LIB_PATH = os.environ['MDPP_LIB']
from test_gpu_parity import _oracle_for, _venv
cfg = {'state_space_type': 'continuous', 'action_space_type': 'continuous', 'state_space_dim': 14, 'action_space_dim': 14, 'transition_dynamics_order': 3, 'inertia': 2.0, 'time_unit': 1.0, 'state_space_max': 6.0, 'action_space_max': 1, 'delay': 1, 'seed': 249, 'reward_scale': 1.0, 'reward_shift': 0.5, 'relevant_indices': [2, 9, 10], 'irrelevant_features': True, 'reward_function': 'move_to_a_point', 'target_point': [-0.44, -0.53, -0.57], 'target_radius': 1.0, 'make_denser': True, 'action_loss_weight': 0.5}
import itertools
for rng, hz, D, order, mode in itertools.product(("philox",), (0, 7), (14, 20), (3, 4), ("same_step",)):
    if True:
        cfg = dict(cfg, transition_dynamics_order=order, state_space_dim=D, action_space_dim=D, relevant_indices=[2, min(9, D - 2), min(10, D - 1)])
        kw = dict(autoreset=mode, max_episode_steps=hz or None)
        if rng == "philox":
            kw.update(rng="philox", philox_seed=77)
        env = _venv(num_envs=512, **kw, **cfg)
        g = np.random.default_rng(1500 + 9)
        acts = (g.uniform(-1, 1, size=(72, 512, D)).astype(np.float32) * np.float32(1.05)).astype(np.float32)
        init = env._obs.cpu().numpy().copy()
        obs, rew, term, trunc = (x.cpu().numpy() for x in env.rollout(torch.as_tensor(acts, device=env.device)))
        bad = []
        for i in range(3, 512, 29):
            o = _oracle_for(env, i)
            if rng == "philox":
                o.set_philox(77, i)
            else:
                o.set_rng(env.seeded_streams[0][i], env.seeded_streams[1][i])
            assert np.array_equal(np.asarray(o.reset()), init[i])
            n = 0
            hist = []
            for t in range(72):
                st, rr, _, d = o.step(acts[t, i])
                n += 1
                tr = bool(hz) and n >= hz
                st_before = np.asarray(st, np.float32).copy()
                if (d or tr) and mode == "same_step":
                    st = o.reset(explicit=False); n = 0
                hist.append((t, int(term[t, i]), int(d), int((np.abs(acts[t, i]) > 1).any())))
                ok_flags = d == bool(term[t, i]) and tr == bool(trunc[t, i])
                ok_obs = np.array_equal(np.asarray(st, np.float32).view(np.uint32), obs[t, i].view(np.uint32))
                if not (ok_flags and ok_obs):
                    rel = [2, 9, 10]
                    dist = float(np.linalg.norm(st_before[rel] - np.array(cfg['target_point'], np.float32)))
                    bad.append((i, t, "flags" if not ok_flags else "obs", d, bool(term[t, i]), tr, bool(trunc[t, i]), round(dist, 6), "reward dev/ora", float(rew[t, i]), float(rr),
                                "obs equal" , ok_obs, "dev obs", obs[t, i][:4].tolist(), "ora next", st_before[:4].tolist(), "status", int(env.status()[i]), ))
                    break
        print(rng, hz, D, order, mode, env.rollout_kernel_name(72), "mismatching envs", len(bad), [b[:8] for b in bad[:3]])
        env.close()
