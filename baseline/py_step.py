"""Pure-Python / numpy restatement of the reference's per-env `step()` and `reset()` — the CPU
baseline leg of the measurement (Leg A/B).

This is what "the reference's pure-Python CPU step()" costs, restated so that it can travel to the
GPU box (the reference's files cannot): ONE Python object per env instance, the same per-step
operation sequence as `RLToyEnv.step` — Python lists for the augmented state and the reward
buffer, a dict lookup of the rewardable sequence, `Generator.choice(p=...)` / `Generator.normal`
draws from the same generators in the same order, `scipy.special.factorial` + numpy ufuncs for
the integrator, `np.linalg.norm` for distances, `Box.contains`-style numpy tests.  What is left
out is only what produces no result: the reference's log-string building and its episode
statistics counters (the speed ratio reference : this file is measured in the build container by
tools/refgen/bench_reference.py and committed as profiles/py_baseline_ratio.json).

It is NOT part of the product (nothing under mdp_playground_amd/ imports it) and NOT the oracle:
tests/test_py_baseline.py pins it to the reference-generated goldens, bench.py times it.

Reference lines followed (/root/reference/mdp_playground/envs/rl_toy_env.py):
  step :1992-2125, transition_function :1577-1725, reward_function :1782-1990, reset :2217-2377,
  DiscreteExtended.sample spaces/discrete_extended.py:11-23, gymnasium Box.sample / Box.contains.
Covers discrete envs (incl. irrelevant_features, both noises, reward matrices) and continuous
`move_to_a_point` envs (any order, both noises, terminal hypercubes): the BASELINE configs.
"""
from __future__ import annotations

import numpy as np
import scipy.special

NAN = float("nan")


def _gen(words_or_gen):
    """A numpy Generator from either a Generator or the 6 PCG64 state words the device uses."""
    if isinstance(words_or_gen, np.random.Generator):
        return words_or_gen
    w = [int(x) for x in words_or_gen]
    bg = np.random.PCG64(0)
    bg.state = {"bit_generator": "PCG64",
                "state": {"state": (w[1] << 64) | w[0], "inc": (w[3] << 64) | w[2]},
                "has_uint32": w[4], "uinteger": w[5]}
    return np.random.Generator(bg)


class PyDiscreteEnv:
    """One discrete RLToyEnv instance (tables handed in, e.g. from mdp_playground_amd.mdp.build_mdp)."""

    def __init__(self, P, rewardable_sequences, terminal_states, init_dist, sequence_length=1, delay=0,
                 reward_every_n_steps=1, transition_noise=None, reward_noise=None, reward_scale=1.0,
                 reward_shift=0.0, term_state_reward=0.0, env_rng=None, space_rng=None,
                 P_irr=None, init_dist_irr=None, space_irr_rng=None, reward_matrix=None):
        self.P = np.asarray(P)
        self.S, self.A = self.P.shape
        self.rewardable_sequences = rewardable_sequences
        self.terminal_states = np.asarray(terminal_states)
        self.init_dist = np.asarray(init_dist, dtype=np.float64)
        self.sequence_length, self.delay = int(sequence_length), int(delay)
        self.augmented_state_length = self.sequence_length + self.delay + 1
        self.reward_every_n_steps = int(reward_every_n_steps)
        self.transition_noise = transition_noise
        self.reward_noise = reward_noise
        self.reward_scale, self.reward_shift = reward_scale, reward_shift
        self.term_state_reward = term_state_reward
        self.np_random = _gen(env_rng)
        self.space_rng = _gen(space_rng)
        self.irrelevant = P_irr is not None
        if self.irrelevant:
            self.P_irr = np.asarray(P_irr)
            self.S_irr = self.P_irr.shape[0]
            self.init_dist_irr = np.asarray(init_dist_irr, dtype=np.float64)
            self.space_irr_rng = _gen(space_irr_rng)
        self.reward_matrix = None if reward_matrix is None else np.asarray(reward_matrix)
        self.reward_buffer = []
        self.augmented_state = []
        self.total_transitions_episode = 0

    def _sample(self, rng, n, prob):
        sampled = np.squeeze(rng.choice(n, size=1, p=prob, replace=True))
        return int(sampled) if sampled.shape == () else sampled

    def reset(self):
        self.reward_buffer = [0.0] * self.delay
        rel = self.np_random.choice(self.S, p=self.init_dist)
        self.curr_state = rel
        if self.irrelevant:
            irr = self.np_random.choice(self.S_irr, p=self.init_dist_irr)
            self.curr_state = (rel, irr)
        self.augmented_state = [NAN for _ in range(self.augmented_state_length - 1)]
        self.augmented_state.append(rel)
        self.total_transitions_episode = 0
        return np.int64(self.curr_state)

    def _noisy(self, rng, n, nxt):
        probs = np.ones(shape=(n,)) * self.transition_noise / (n - 1)
        probs[nxt] = 1 - self.transition_noise
        return self._sample(rng, n, probs)

    def step(self, action):
        if self.irrelevant:
            state, act, state_irr, act_irr = self.curr_state[0], action[0], self.curr_state[1], action[1]
        else:
            state, act = self.curr_state, action
        next_state = self.P[state, act]
        if self.transition_noise:
            next_state = self._noisy(self.space_rng, self.S, next_state)
        del self.augmented_state[0]
        self.augmented_state.append(next_state)
        self.total_transitions_episode += 1
        # ---- reward
        reward = 0.0
        if self.reward_matrix is not None:
            reward = self.reward_matrix[self.augmented_state[-2], act]
        elif np.isnan(self.augmented_state[self.delay]):
            pass
        else:
            sub_seq = tuple(self.augmented_state[1 + self.delay:self.augmented_state_length])
            if sub_seq in self.rewardable_sequences:
                reward = self.rewardable_sequences[sub_seq]
        self.reward_buffer.append(reward)
        reward = self.reward_buffer[0]
        del self.reward_buffer[0]
        if self.total_transitions_episode % self.reward_every_n_steps != 0:
            reward = 0.0
        noise = self.np_random.normal(0, self.reward_noise) if self.reward_noise is not None else 0
        reward += noise
        reward *= self.reward_scale
        reward += self.reward_shift
        # ---- irrelevant sub-space
        if self.irrelevant:
            nxt_irr = self.P_irr[state_irr, act_irr]
            if self.transition_noise:
                nxt_irr = self._noisy(self.space_irr_rng, self.S_irr, nxt_irr)
            next_state = (next_state, nxt_irr)
        self.curr_state = np.int64(next_state)
        done = self.augmented_state[-1] in self.terminal_states
        if done:
            reward += self.term_state_reward * self.reward_scale
        return self.curr_state, reward, done, False


class PyContinuousEnv:
    """One continuous `move_to_a_point` RLToyEnv instance."""

    def __init__(self, D, relevant_indices, order=1, inertia=1.0, time_unit=1.0, state_space_max=np.inf,
                 action_space_max=np.inf, target_point=None, target_radius=0.05, make_denser=True,
                 action_loss_weight=0.0, transition_noise=None, reward_noise=None, delay=0,
                 reward_every_n_steps=1, reward_scale=1.0, reward_shift=0.0, term_state_reward=0.0,
                 box_lo=None, box_hi=None, env_rng=None, space_rng=None):
        self.D, self.rel = int(D), list(relevant_indices)
        self.order, self.inertia, self.time_unit = int(order), inertia, time_unit
        self.state_space_max, self.action_space_max = state_space_max, action_space_max
        f32 = np.float32
        self.s_low, self.s_high = np.full(D, -state_space_max, f32), np.full(D, state_space_max, f32)
        self.a_low, self.a_high = np.full(D, -action_space_max, f32), np.full(D, action_space_max, f32)
        # (no target_point: float64 zeros of length state_space_dim, rl_toy_env.py:652-654)
        self.target_point = np.zeros(shape=(D,)) if target_point is None else np.array(target_point, dtype=f32)
        self.target_radius, self.make_denser = target_radius, make_denser
        self.action_loss_weight = action_loss_weight
        self.transition_noise, self.reward_noise = transition_noise, reward_noise
        self.delay, self.reward_every_n_steps = int(delay), int(reward_every_n_steps)
        self.augmented_state_length = 1 + self.delay + 1          # sequence_length 1
        self.reward_scale, self.reward_shift = reward_scale, reward_shift
        self.term_state_reward = term_state_reward
        self.boxes = [] if box_lo is None else [(np.asarray(lo, f32), np.asarray(hi, f32))
                                                for lo, hi in zip(box_lo, box_hi)]
        self.np_random, self.space_rng = _gen(env_rng), _gen(space_rng)

    @staticmethod
    def _contains(x, low, high):
        return bool(np.can_cast(x.dtype, np.float32) and x.shape == low.shape
                    and np.all(x >= low) and np.all(x <= high))

    def is_terminal_state(self, s):
        return bool(np.any([self._contains(s[self.rel], lo, hi) for lo, hi in self.boxes])) if self.boxes else False

    def _sample_state(self):
        if np.isfinite(self.state_space_max):
            sample = self.space_rng.uniform(low=self.s_low, high=self.s_high, size=(self.D,))
        else:
            sample = self.space_rng.normal(size=(self.D,))
        return sample.astype(np.float32)

    def reset(self):
        self.reward_buffer = [0.0] * self.delay
        while True:
            self.curr_state = self._sample_state()
            if not self.is_terminal_state(self.curr_state):
                break
        zero = np.array([0.0] * self.D, dtype=np.float32)
        self.state_derivatives = [zero.copy() for _ in range(self.order + 1)]
        self.state_derivatives[0] = self.curr_state.copy()
        self.augmented_state = [[NAN] * self.D for _ in range(self.augmented_state_length - 1)]
        self.augmented_state.append(self.curr_state.copy())
        self.reached_terminal = False
        self.total_transitions_episode = 0
        return self.curr_state

    def step(self, action):
        state = self.curr_state
        # ---- transition
        if self._contains(action, self.a_low, self.a_high):
            self.state_derivatives[-1] = action / self.inertia
            factorial_array = scipy.special.factorial(np.arange(1, self.order + 1))
            for i in range(self.order):
                for j in range(self.order - i):
                    self.state_derivatives[i] += (self.state_derivatives[i + j + 1]
                                                  * (self.time_unit ** (j + 1)) / factorial_array[j])
            next_state = self.state_derivatives[0].copy()
        else:
            next_state = state
        noise = (self.np_random.normal(0, self.transition_noise, state.shape)
                 if self.transition_noise is not None else np.zeros(self.D))
        next_state += noise
        if not self._contains(next_state, self.s_low, self.s_high):
            next_state = np.clip(next_state, -self.state_space_max, self.state_space_max)
            zero = np.array([0.0] * self.D, dtype=np.float32)
            self.state_derivatives = [zero.copy() for _ in range(self.order + 1)]
            self.state_derivatives[0] = next_state.copy()
        next_rel = np.array(next_state, dtype=np.float32)[self.rel]
        if np.linalg.norm(next_rel - self.target_point) < self.target_radius:
            self.reached_terminal = True
        del self.augmented_state[0]
        self.augmented_state.append(next_state.copy())
        self.total_transitions_episode += 1
        # ---- reward
        reward = 0.0
        sc = self.augmented_state
        if np.isnan(sc[self.delay][0]):
            pass
        else:
            if self.make_denser:
                old_rel = np.array(sc, dtype=np.float32)[-2, self.rel]
                new_rel = np.array(sc, dtype=np.float32)[-1, self.rel]
                reward = -np.linalg.norm(new_rel - self.target_point)
                reward += np.linalg.norm(old_rel - self.target_point)
            else:
                new_rel = np.array(sc, dtype=np.float32)[-1, self.rel]
                if np.linalg.norm(new_rel - self.target_point) < self.target_radius:
                    reward = 1.0
            reward -= self.action_loss_weight * np.linalg.norm(np.array(action, dtype=np.float32))
        self.reward_buffer.append(reward)
        reward = self.reward_buffer[0]
        del self.reward_buffer[0]
        if self.total_transitions_episode % self.reward_every_n_steps != 0:
            reward = 0.0
        rnoise = self.np_random.normal(0, self.reward_noise) if self.reward_noise is not None else 0
        reward += rnoise
        reward *= self.reward_scale
        reward += self.reward_shift
        self.curr_state = np.float32(next_state)
        done = self.is_terminal_state(self.augmented_state[-1]) or self.reached_terminal
        if done:
            reward += self.term_state_reward * self.reward_scale
        return self.curr_state, reward, done, False


def from_mdp(m, env_rng, space_rng, space_irr_rng=None):
    """A baseline env for an MDP built by mdp_playground_amd.mdp.build_mdp (host-side tables)."""
    if m.kind == "discrete":
        return PyDiscreteEnv(m.P, m.rewardable_sequences, m.terminal_states, m.init_dist,
                             sequence_length=m.sequence_length, delay=m.delay,
                             reward_every_n_steps=m.reward_every_n_steps, transition_noise=m.transition_noise,
                             reward_noise=m.reward_noise, reward_scale=m.reward_scale, reward_shift=m.reward_shift,
                             term_state_reward=m.term_state_reward, env_rng=env_rng, space_rng=space_rng,
                             P_irr=m.P_irr if m.irrelevant else None,
                             init_dist_irr=m.init_dist_irr if m.irrelevant else None,
                             space_irr_rng=space_irr_rng, reward_matrix=m.reward_matrix)
    if m.kind == "continuous" and m.reward_function == "move_to_a_point":
        return PyContinuousEnv(m.D, m.relevant_indices, order=m.order, inertia=m.inertia, time_unit=m.time_unit,
                               state_space_max=m.state_space_max, action_space_max=m.action_space_max,
                               target_point=None if m.target_default else m.target_point, target_radius=m.target_radius,
                               make_denser=m.make_denser, action_loss_weight=m.action_loss_weight,
                               transition_noise=m.transition_noise, reward_noise=m.reward_noise, delay=m.delay,
                               reward_every_n_steps=m.reward_every_n_steps, reward_scale=m.reward_scale,
                               reward_shift=m.reward_shift, term_state_reward=m.term_state_reward,
                               box_lo=m.box_lo, box_hi=m.box_hi, env_rng=env_rng, space_rng=space_rng)
    raise NotImplementedError(f"no pure-Python baseline for {m.kind} / {getattr(m, 'reward_function', '')}")
