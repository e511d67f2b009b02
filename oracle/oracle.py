"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/_build/libmdpp_oracle.so (the scalar C restatement of
RLToyEnv.step()/reset(); see mdpp_oracle.h for the reference file:line map).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (mdp_playground_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDPP_ORACLE_SANITIZE=1: load the AddressSanitizer + UBSan build instead (the process must have libasan preloaded:
# tests/test_oracle_sanitized.py starts such a child)
_SANITIZE = os.environ.get("MDPP_ORACLE_SANITIZE", "") == "1"
_SO = os.path.join(_HERE, "_build", "libmdpp_oracle_san.so" if _SANITIZE else "libmdpp_oracle.so")


def build(force=False, sanitize=None):
    """Compile the restatement (gcc; `make -C oracle`).  sanitize=True: the -fsanitize=address,undefined build,
    oracle/_build/libmdpp_oracle_san.so (`make sanitize`).  Returns the path of the library built."""
    sanitize = _SANITIZE if sanitize is None else sanitize
    so = os.path.join(_HERE, "_build", "libmdpp_oracle_san.so" if sanitize else "libmdpp_oracle.so")
    srcs = ["mdpp_oracle.c", "np_random.c", "mdpp_oracle.h", "np_random.h",
            "np_ziggurat_tables.inc", "Makefile"]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + (["sanitize"] if sanitize else []))
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.ora_d_create.restype = vp
        L.ora_d_create.argtypes = [i32] * 5 + [i32, f64, i32, f64, f64, f64, f64] + [vp] * 4
        L.ora_d_destroy.argtypes = [vp]
        L.ora_d_set_rng.argtypes = [vp, vp, vp]
        L.ora_d_get_rng.argtypes = [vp, vp, vp]
        L.ora_d_reset.restype = C.c_int64
        L.ora_d_reset.argtypes = [vp]
        L.ora_d_step.argtypes = [vp, i32, vp, vp, vp]
        L.ora_d_rollout.argtypes = [vp, i32] + [vp] * 6
        L.ora_d_set_irrelevant.argtypes = [vp, i32, i32, vp, vp]
        L.ora_d_set_reward_matrix.argtypes = [vp, vp]
        L.ora_d_set_rng_irr.argtypes = [vp, vp]
        L.ora_d_get_rng_irr.argtypes = [vp, vp]
        L.ora_d_reset2.argtypes = [vp, vp]
        L.ora_d_step2.argtypes = [vp, i32, i32, vp, vp, vp]
        L.ora_d_rollout2.argtypes = [vp, i32] + [vp] * 6
        L.ora_g_create.restype = vp
        L.ora_g_create.argtypes = [i32, vp, vp, i32, i32, f64, i32, f64, i32, f64, f64, f64]
        L.ora_g_destroy.argtypes = [vp]
        L.ora_g_set_rng.argtypes = [vp, vp, vp, vp]
        L.ora_g_get_rng.argtypes = [vp, vp, vp, vp]
        L.ora_g_set_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_g_philox_explicit_reset.argtypes = [vp]
        L.ora_g_reset.argtypes = [vp, vp]
        L.ora_g_step.argtypes = [vp, vp, vp, vp, vp]
        L.ora_g_rollout.argtypes = [vp, i32] + [vp] * 6
        L.ora_c_create.restype = vp
        L.ora_c_create.argtypes = ([i32, i32, vp, i32, f64, f64, f64, f64, vp, f64, i32, f64,
                                    i32, f64, i32, f64, i32, i32, f64, f64, f64, i32, vp, vp])
        L.ora_c_destroy.argtypes = [vp]
        L.ora_c_set_image_quirk.argtypes = [vp, i32]
        L.ora_c_set_target64.argtypes = [vp]
        L.ora_c_get_stats.argtypes = [vp, vp, vp]
        L.ora_d_get_stats.argtypes = [vp, vp, vp]
        L.ora_g_get_stats.argtypes = [vp, vp, vp]
        L.ora_c_set_line_reward.argtypes = [vp, i32, LINE_FIT_FN]
        L.ora_ig_render.argtypes = [i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp]
        L.ora_ic_render.argtypes = [i32, i32, i32, vp, i32, vp, C.c_float, vp, i32, vp, vp, vp]
        L.ora_c_set_rng.argtypes = [vp, vp, vp]
        L.ora_c_get_rng.argtypes = [vp, vp, vp]
        L.ora_c_reset.argtypes = [vp, vp]
        L.ora_c_step.argtypes = [vp] * 6
        L.ora_c_get_derivs.argtypes = [vp, vp]
        L.ora_c_rollout.argtypes = [vp, i32] + [vp] * 6
        L.ora_d_set_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_c_set_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_d_philox_explicit_reset.argtypes = [vp]
        L.ora_c_philox_explicit_reset.argtypes = [vp]
        L.ora_i_draw.argtypes = [vp] * 7
        L.ora_i_draw_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, i32, vp]
        L.ora_i_rotate_flip_transpose.argtypes = [i32, i32, vp, i32, i32, vp]
        for name in ("np_next64", "np_next32"):
            getattr(L, name).argtypes = [vp]
        L.np_next64.restype = C.c_uint64
        L.np_next32.restype = C.c_uint32
        L.np_random.restype = f64
        L.np_random.argtypes = [vp]
        L.np_standard_normal.restype = f64
        L.np_standard_normal.argtypes = [vp]
        L.np_integers.restype = C.c_int64
        L.np_integers.argtypes = [vp, C.c_int64, C.c_int64]
        L.ora_p_create.restype = vp
        L.ora_p_create.argtypes = [i32] * 6 + [f64, i32, f64, f64, f64, f64] + [i32] * 7
        L.ora_p_destroy.argtypes = [vp]
        L.ora_p_set_rng.argtypes = [vp, vp]
        L.ora_p_get_rng.argtypes = [vp, vp]
        L.ora_p_set_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_p_reset.argtypes = [vp, vp, vp]
        L.ora_p_action.argtypes = [vp, i32]
        L.ora_p_action.restype = i32
        L.ora_p_step.argtypes = [vp, vp, f64, i32, vp, vp]
        L.ora_np_pairwise_sum.argtypes = [vp, i32]
        L.ora_np_pairwise_sum.restype = f64
        L.np_philox_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, i32, i32, vp]
        L.np_philox_box_muller.argtypes = [C.c_uint32, C.c_uint32, vp, vp]
        L.np_philox_tick_word.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.np_philox_tick_word.restype = C.c_uint32
        L.np_philox_pnoise_state.argtypes = [C.c_uint32, C.c_double, C.c_int, C.c_int]
        L.np_philox_pnoise_state.restype = C.c_int
        L.np_philox_tick_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.np_philox_tick_normal.restype = C.c_float
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pcg_words(gen):
    """6 x uint64 words of a numpy Generator(PCG64) state: s_lo, s_hi, inc_lo, inc_hi, has32, u32."""
    st = gen.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s & m, s >> 64, inc & m, inc >> 64, st["has_uint32"], st["uinteger"]],
                    dtype=np.uint64)


class NpPCG64(C.Structure):
    """Mirror of np_pcg64 for driving the RNG primitives directly from tests."""
    _fields_ = [("s_lo", C.c_uint64), ("s_hi", C.c_uint64), ("inc_lo", C.c_uint64),
                ("inc_hi", C.c_uint64), ("has32", C.c_uint32), ("u32", C.c_uint32),
                ("philox", C.c_uint32), ("k0", C.c_uint32), ("k1", C.c_uint32),
                ("c0", C.c_uint32), ("c1", C.c_uint32), ("c2", C.c_uint32), ("c3", C.c_uint32),
                ("spare_lo", C.c_uint32), ("spare_hi", C.c_uint32), ("have_spare", C.c_uint32),
                ("z_spare", C.c_float), ("have_z", C.c_uint32)]

    @classmethod
    def from_words(cls, w):
        w = [int(x) for x in w]
        return cls(w[0], w[1], w[2], w[3], w[4], w[5])


class PostOracle:
    """One GymEnvWrapper instance's post-processing (gym_env_wrapper.py:350-439, :441-486, :523-618) of an
    inner env's (obs, reward, done); config keys as the wrapper's."""

    def __init__(self, state_space_type, n_actions=0, obs_dim=0, obs_dtype=np.float32, delay=0,
                 transition_noise=None, reward_noise=None, reward_scale=1.0, reward_shift=0.0, term_state_reward=0.0,
                 image_shape=None, image_transforms=None, image_padding=20, image_sh_quant=1):
        self.cont = state_space_type == "continuous"
        self.obs_dtype = np.dtype(obs_dtype)
        self.obs_dim = int(obs_dim)
        self.image = image_transforms is not None and bool(image_transforms)
        self.shape = tuple(image_shape) if self.image else None
        H, W, Cc = self.shape if self.image else (0, 0, 0)
        self.pad = int(image_padding)
        self.h = lib().ora_p_create(int(self.cont), int(n_actions), self.obs_dim, int(self.obs_dtype == np.float64),
                                    int(delay), int(transition_noise is not None), float(transition_noise or 0.0),
                                    int(reward_noise is not None), float(reward_noise or 0.0), float(reward_scale),
                                    float(reward_shift), float(term_state_reward), int(self.image), H, W, Cc, self.pad,
                                    int(self.image and "shift" in image_transforms), int(image_sh_quant or 1))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_p_destroy(self.h)
            self.h = None

    def set_rng(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        lib().ora_p_set_rng(self.h, _p(w))

    def get_rng(self):
        w = np.zeros(6, np.uint64)
        lib().ora_p_get_rng(self.h, _p(w))
        return w

    def set_philox(self, seed, env_id, tick=0, reset_tick=0):
        lib().ora_p_set_philox(self.h, int(seed), int(env_id), int(tick), int(reset_tick))

    def _img_out(self):
        H, W, Cc = self.shape
        return np.zeros((W + 2 * self.pad, H + 2 * self.pad, Cc), np.uint8)

    def reset(self, obs=None):
        if not self.image:
            lib().ora_p_reset(self.h, None, None)
            return obs
        src = np.ascontiguousarray(obs, dtype=np.uint8)
        out = self._img_out()
        lib().ora_p_reset(self.h, _p(src), _p(out))
        return out

    def action(self, a):
        return int(lib().ora_p_action(self.h, int(a)))

    def step(self, obs, reward, done):
        r = C.c_double()
        if self.cont:
            src = np.ascontiguousarray(obs, dtype=self.obs_dtype)
            out = np.zeros_like(src)
            lib().ora_p_step(self.h, _p(src), float(reward), int(bool(done)), _p(out), C.byref(r))
            return out, r.value
        if self.image:
            src = np.ascontiguousarray(obs, dtype=np.uint8)
            out = self._img_out()
            lib().ora_p_step(self.h, _p(src), float(reward), int(bool(done)), _p(out), C.byref(r))
            return out, r.value
        lib().ora_p_step(self.h, None, float(reward), int(bool(done)), None, C.byref(r))
        return obs, r.value


def philox_normals(seed, env0, tick, stream, n_envs, n_per_env):
    """[n_envs, n_per_env] standard normals of the Philox-mode streams (seed, env0 + e, tick, stream)."""
    out = np.zeros((n_envs, n_per_env), np.float64)
    lib().np_philox_normals(int(seed), int(env0), int(tick), int(stream), n_envs, n_per_env, _p(out))
    return out


def philox_tick_word(seed, env, tick, stream):
    """Philox mode, discrete envs: the 32-bit word tick `tick` draws from stream `stream` (np_random.c np_philox_tick_word)."""
    return int(lib().np_philox_tick_word(int(seed), int(env), int(tick), int(stream)))


def philox_pnoise_state(w, p, S, nxt):
    """... and the state a step lands in when the table says `nxt` (np_philox_pnoise_state)."""
    return int(lib().np_philox_pnoise_state(int(w), float(p), int(S), int(nxt)))


def philox_tick_normal(seed, env, tick, stream):
    """... and its reward-noise normal (float32)."""
    return float(lib().np_philox_tick_normal(int(seed), int(env), int(tick), int(stream)))


def rtable_from_sequences(S, L, keys, vals):
    """Dense float64[S**L] reward table; key = sum seq[i] * S**(L-1-i)."""
    t = np.zeros(S ** L, dtype=np.float64)
    for seq, v in zip(keys, vals):
        k = 0
        for s in seq:
            k = k * S + int(s)
        t[k] = v
    return t


class DiscreteOracle:
    def __init__(self, S, A, L, delay, every_n, P, rtable, terminal_states, init_dist,
                 transition_noise=None, reward_noise=None, reward_scale=1.0,
                 reward_shift=0.0, term_state_reward=0.0):
        self.S, self.A, self.L = S, A, L
        self._P = np.ascontiguousarray(P, dtype=np.int32).reshape(S, A)
        self._rt = np.ascontiguousarray(rtable, dtype=np.float64)
        assert self._rt.size == S ** L
        it = np.zeros(S, dtype=np.uint8)
        it[np.asarray(terminal_states, dtype=np.int64)] = 1
        self._it = it
        self._id = np.ascontiguousarray(init_dist, dtype=np.float64)
        has_p = bool(transition_noise)                 # rl_toy_env.py:1604 truthiness
        has_r = reward_noise is not None               # :398-405 lambda exists even for std 0
        self.h = lib().ora_d_create(S, A, L, delay, every_n, int(has_p),
                                    float(transition_noise or 0.0), int(has_r),
                                    float(reward_noise or 0.0), float(reward_scale),
                                    float(reward_shift), float(term_state_reward),
                                    _p(self._P), _p(self._rt), _p(self._it), _p(self._id))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_d_destroy(self.h)
            self.h = None

    def set_rng(self, env_words, space_words):
        a = np.ascontiguousarray(env_words, dtype=np.uint64)
        b = np.ascontiguousarray(space_words, dtype=np.uint64)
        lib().ora_d_set_rng(self.h, _p(a), _p(b))

    def get_rng(self):
        a = np.zeros(6, np.uint64)
        b = np.zeros(6, np.uint64)
        lib().ora_d_get_rng(self.h, _p(a), _p(b))
        return a, b

    def get_stats(self):
        """(current [3], last [4]): total_abs_noise_in_reward_episode, total_reward_episode,
        total_noisy_transitions_episode of the running episode / of the episode the latest reset() ended (+ its
        total_transitions_episode), rl_toy_env.py:2231-2247."""
        cur, last = np.zeros(3), np.zeros(4)
        lib().ora_d_get_stats(self.h, _p(cur), _p(last))
        return cur, last

    def set_philox(self, seed, env_id, tick=0, reset_tick=0):
        self._philox = True
        lib().ora_d_set_philox(self.h, int(seed), int(env_id), int(tick), int(reset_tick))

    def set_reward_matrix(self, R):
        """use_custom_mdp with reward_function given as an S x A array (rl_toy_env.py:1259-1267)."""
        R = np.ascontiguousarray(R, dtype=np.float64)
        assert R.shape == (self.S, self.A)
        lib().ora_d_set_reward_matrix(self.h, _p(R))

    # ---- irrelevant sub-space (Tuple observations / actions): states and actions become pairs
    def set_irrelevant(self, P_irr, init_dist_irr):
        P1 = np.ascontiguousarray(P_irr, dtype=np.int32)
        d1 = np.ascontiguousarray(init_dist_irr, dtype=np.float64)
        self._irr = True
        lib().ora_d_set_irrelevant(self.h, P1.shape[0], P1.shape[1], _p(P1), _p(d1))

    def set_rng_irr(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        lib().ora_d_set_rng_irr(self.h, _p(w))

    def get_rng_irr(self):
        w = np.zeros(6, np.uint64)
        lib().ora_d_get_rng_irr(self.h, _p(w))
        return w

    def reset(self, explicit=True):
        """explicit=True mirrors a reset() call of its own (mdpp_reset); False an in-step reset."""
        if getattr(self, "_philox", False) and explicit:
            lib().ora_d_philox_explicit_reset(self.h)
        if getattr(self, "_irr", False):
            o = np.zeros(2, np.int64)
            lib().ora_d_reset2(self.h, _p(o))
            return int(o[0]), int(o[1])
        return int(lib().ora_d_reset(self.h))

    def step(self, action):
        r = C.c_double()
        d = C.c_uint8()
        if getattr(self, "_irr", False):
            o = np.zeros(2, np.int64)
            lib().ora_d_step2(self.h, int(action[0]), int(action[1]), _p(o), C.byref(r), C.byref(d))
            return (int(o[0]), int(o[1])), r.value, bool(d.value)
        o = C.c_int64()
        lib().ora_d_step(self.h, int(action), C.byref(o), C.byref(r), C.byref(d))
        return o.value, r.value, bool(d.value)

    def rollout(self, actions, reset_after=None):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        T = actions.shape[0]
        ra = None if reset_after is None else np.ascontiguousarray(reset_after, dtype=np.uint8)
        if getattr(self, "_irr", False):
            obs = np.zeros((T, 2), np.int64)
            rew = np.zeros(T, np.float64)
            done = np.zeros(T, np.uint8)
            ro = np.zeros((T, 2), np.int64)
            lib().ora_d_rollout2(self.h, T, _p(actions), _p(ra), _p(obs), _p(rew), _p(done), _p(ro))
            return obs, rew, done.astype(bool), ro
        obs = np.zeros(T, np.int64)
        rew = np.zeros(T, np.float64)
        done = np.zeros(T, np.uint8)
        ro = np.zeros(T, np.int64)
        lib().ora_d_rollout(self.h, T, _p(actions), _p(ra), _p(obs), _p(rew), _p(done), _p(ro))
        return obs, rew, done.astype(bool), ro


class GridOracle:
    def __init__(self, grid_shape, target_point, make_denser, transition_noise=None, reward_noise=None,
                 every_n=1, reward_scale=1.0, reward_shift=0.0, term_state_reward=0.0):
        self.G = len(grid_shape)
        sh = np.ascontiguousarray(grid_shape, dtype=np.int32)
        tg = np.ascontiguousarray(target_point, dtype=np.int32)
        self.h = lib().ora_g_create(self.G, _p(sh), _p(tg), int(bool(make_denser)),
                                    int(bool(transition_noise)), float(transition_noise or 0.0),
                                    int(reward_noise is not None), float(reward_noise or 0.0),
                                    int(every_n), float(reward_scale), float(reward_shift),
                                    float(term_state_reward))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_g_destroy(self.h)
            self.h = None

    def set_rng(self, env_words, space_words, action_words):
        w = [np.ascontiguousarray(x, dtype=np.uint64) for x in (env_words, space_words, action_words)]
        lib().ora_g_set_rng(self.h, _p(w[0]), _p(w[1]), _p(w[2]))

    def get_rng(self):
        w = [np.zeros(6, np.uint64) for _ in range(3)]
        lib().ora_g_get_rng(self.h, _p(w[0]), _p(w[1]), _p(w[2]))
        return w

    def get_stats(self):
        cur, last = np.zeros(3), np.zeros(4)
        lib().ora_g_get_stats(self.h, _p(cur), _p(last))
        return cur, last

    def set_philox(self, seed, env_id, tick=0, reset_tick=0):
        self._philox = True
        lib().ora_g_set_philox(self.h, int(seed), int(env_id), int(tick), int(reset_tick))

    def reset(self, explicit=True):
        if getattr(self, "_philox", False) and explicit:
            lib().ora_g_philox_explicit_reset(self.h)
        o = np.zeros(self.G, np.int64)
        lib().ora_g_reset(self.h, _p(o))
        return o

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        o = np.zeros(self.G, np.int64)
        r = C.c_double()
        d = C.c_uint8()
        lib().ora_g_step(self.h, _p(a), _p(o), C.byref(r), C.byref(d))
        return o, r.value, bool(d.value)

    def rollout(self, actions, reset_after=None):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        T = actions.shape[0]
        ra = None if reset_after is None else np.ascontiguousarray(reset_after, dtype=np.uint8)
        obs = np.zeros((T, self.G), np.int64)
        rew = np.zeros(T, np.float64)
        done = np.zeros(T, np.uint8)
        ro = np.zeros((T, self.G), np.int64)
        lib().ora_g_rollout(self.h, T, _p(actions), _p(ra), _p(obs), _p(rew), _p(done), _p(ro))
        return obs, rew, done.astype(bool), ro


# what the reference's move_along_a_line reward gets from numpy / LAPACK (rl_toy_env.py:1865-1871)
LINE_FIT_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_double),
                          C.POINTER(C.c_double))


class ContinuousOracle:
    def __init__(self, D, relevant_indices, order, inertia, time_unit, state_space_max,
                 action_space_max, target_point, target_radius, make_denser,
                 action_loss_weight=0.0, transition_noise=None, reward_noise=None, delay=0,
                 every_n=1, reward_scale=1.0, reward_shift=0.0, term_state_reward=0.0,
                 box_lo=None, box_hi=None):
        self.D, self.order = D, order
        rel = np.ascontiguousarray(relevant_indices, dtype=np.int32)
        tgt = np.ascontiguousarray(target_point, dtype=np.float32)
        nb = 0 if box_lo is None else len(box_lo)
        blo = None if nb == 0 else np.ascontiguousarray(box_lo, dtype=np.float32)
        bhi = None if nb == 0 else np.ascontiguousarray(box_hi, dtype=np.float32)
        self._keep = (rel, tgt, blo, bhi)
        self.h = lib().ora_c_create(
            D, len(rel), _p(rel), order, float(inertia), float(time_unit),
            float(state_space_max), float(action_space_max), _p(tgt), float(target_radius),
            int(bool(make_denser)), float(action_loss_weight),
            int(transition_noise is not None), float(transition_noise or 0.0),
            int(reward_noise is not None), float(reward_noise or 0.0),
            int(delay), int(every_n), float(reward_scale), float(reward_shift),
            float(term_state_reward), nb, _p(blo), _p(bhi))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_c_destroy(self.h)
            self.h = None

    def set_image_quirk(self, on=True):
        """image_representations=True: every step clips and zeroes the derivatives (mdpp_oracle.c C4)."""
        lib().ora_c_set_image_quirk(self.h, int(on))

    def set_target64(self):
        """No target_point in the config: the reference's float64 zeros of length state_space_dim (:652-654), i.e.
        float64 distances, target latch and dense reward (mdpp_oracle.c ora_c_set_target64)."""
        lib().ora_c_set_target64(self.h)

    def set_line_reward(self, sequence_length, delay=0):
        """reward_function='move_along_a_line': the float32 mean and the first right-singular vector
        come from numpy exactly as the reference computes them (rl_toy_env.py:1865-1871: an
        (augmented_state_length, D) float32 array, rows [1 + delay:], columns relevant_indices, so
        that the slice has the reference's memory layout and numpy sums it the same way)."""
        rel = [int(x) for x in self._keep[0]]
        L, D = int(sequence_length), self.D

        def fit(pts, L_, D_, mean_out, v0_out):
            states = np.ctypeslib.as_array(pts, shape=(L_ * D_,)).reshape(L_, D_)
            considered = [np.full(D_, np.nan, np.float32)] * (1 + delay) + [states[k].copy() for k in range(L_)]
            data_ = np.array(considered, dtype=np.float32)[1 + delay:1 + delay + L_, rel]
            data_mean = data_.mean(axis=0)
            uu, dd, vv = np.linalg.svd(data_ - data_mean)
            for j in range(len(rel)):
                mean_out[j] = float(data_mean[j])
                v0_out[j] = float(vv[0][j])
        self._line_cb = LINE_FIT_FN(fit)             # keep the thunk alive
        lib().ora_c_set_line_reward(self.h, L, self._line_cb)

    def set_rng(self, env_words, space_words):
        a = np.ascontiguousarray(env_words, dtype=np.uint64)
        b = np.ascontiguousarray(space_words, dtype=np.uint64)
        lib().ora_c_set_rng(self.h, _p(a), _p(b))

    def get_rng(self):
        a = np.zeros(6, np.uint64)
        b = np.zeros(6, np.uint64)
        lib().ora_c_get_rng(self.h, _p(a), _p(b))
        return a, b

    def get_stats(self):
        """(current [3 + D], last [4 + D]): rows as mdpp_get_episode_stats (include/mdpp.h)."""
        cur, last = np.zeros(3 + self.D), np.zeros(4 + self.D)
        lib().ora_c_get_stats(self.h, _p(cur), _p(last))
        return cur, last

    def set_philox(self, seed, env_id, tick=0, reset_tick=0):
        self._philox = True
        lib().ora_c_set_philox(self.h, int(seed), int(env_id), int(tick), int(reset_tick))

    def reset(self, explicit=True):
        if getattr(self, "_philox", False) and explicit:
            lib().ora_c_philox_explicit_reset(self.h)
        obs = np.zeros(self.D, np.float32)
        lib().ora_c_reset(self.h, _p(obs))
        return obs

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float32)
        obs = np.zeros(self.D, np.float32)
        r = C.c_double()
        is32 = C.c_int()
        d = C.c_uint8()
        lib().ora_c_step(self.h, _p(a), _p(obs), C.byref(r), C.byref(is32), C.byref(d))
        return obs, r.value, bool(is32.value), bool(d.value)

    def derivs(self):
        sd = np.zeros((self.order + 1, self.D), np.float32)
        lib().ora_c_get_derivs(self.h, _p(sd))
        return sd

    def rollout(self, actions, reset_after=None):
        actions = np.ascontiguousarray(actions, dtype=np.float32)
        T = actions.shape[0]
        ra = None if reset_after is None else np.ascontiguousarray(reset_after, dtype=np.uint8)
        obs = np.zeros((T, self.D), np.float32)
        rew = np.zeros(T, np.float64)
        done = np.zeros(T, np.uint8)
        ro = np.zeros((T, self.D), np.float32)
        lib().ora_c_rollout(self.h, T, _p(actions), _p(ra), _p(obs), _p(rew), _p(done), _p(ro))
        return obs, rew, done.astype(bool), ro


class ImageCfg(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("has_scale", C.c_int), ("has_shift", C.c_int),
                ("has_rotate", C.c_int), ("has_flip", C.c_int), ("sh_quant", C.c_int),
                ("ro_quant", C.c_int), ("R0", C.c_int), ("log_min_R", C.c_double),
                ("log_max_R", C.c_double)]


def disc_template(R):
    """(2R+1)^2 raster of Pillow's ellipse with the integer bounding box centre +- R (what
    ImageContinuous draws for the agent and the target, spaces/image_continuous.py:190-207)."""
    import PIL.Image as Image
    import PIL.ImageDraw as ImageDraw
    T = 2 * R + 1
    img = Image.new("L", (T, T), 0)
    ImageDraw.Draw(img).ellipse([(0, 0), (2 * R, 2 * R)], fill=255)
    return (np.array(img) != 0).astype(np.uint8)


def image_continuous_render(W, H, R, state, smax, target, box_lo=None, box_hi=None):
    """uint8 [n_sub * W, H, 3] observation of a continuous state (2 or 4 dims)."""
    state = np.ascontiguousarray(state, dtype=np.float32)
    D = state.shape[0]
    disc = np.ascontiguousarray(disc_template(R))
    tgt = np.ascontiguousarray(target, dtype=np.float32)
    nb = 0 if box_lo is None else len(box_lo)
    lo = None if nb == 0 else np.ascontiguousarray(box_lo, dtype=np.float32)
    hi = None if nb == 0 else np.ascontiguousarray(box_hi, dtype=np.float32)
    out = np.zeros(((2 if D > 2 else 1) * W, H, 3), np.uint8)
    lib().ora_ic_render(W, H, R, _p(disc), D, _p(state), C.c_float(smax), _p(tgt), nb, _p(lo), _p(hi), _p(out))
    return out


def grid_line_mask(W, H, grid_shape):
    """uint8 [n_sub * W, H]: 1 where ImageContinuous draws its white grid lines, made with Pillow's
    draw.line from the reference's own end points (spaces/image_continuous.py:145-165, incl. its use
    of the x-count of the grid for the horizontal spacing)."""
    import PIL.Image as Image
    import PIL.ImageDraw as ImageDraw
    out = []
    for offset in range(0, len(grid_shape), 2):
        img = Image.new("L", (W, H), 0)
        d = ImageDraw.Draw(img)
        for i in range(1, grid_shape[0 + offset] + 1):
            x_ = i * W // grid_shape[0 + offset] - 1
            d.line([(x_, H), (x_, 0)], fill=255)
        for j in range(1, grid_shape[1 + offset]):
            y_ = j * H // grid_shape[0 + offset]
            d.line([(W, y_), (0, y_)], fill=255)
        out.append((np.array(img).T != 0).astype(np.uint8))     # [x][y]
    return np.concatenate(out, axis=0)


def image_grid_render(W, H, R, grid_shape, cells, target, terminal_cells):
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    G = cells.shape[0]
    sh = np.ascontiguousarray(grid_shape, dtype=np.int32)
    tg = np.ascontiguousarray(target, dtype=np.int32)
    tc = np.ascontiguousarray(terminal_cells if terminal_cells is not None else [], dtype=np.int32).reshape(-1, 2)
    disc = np.ascontiguousarray(disc_template(R))
    lines = np.ascontiguousarray(grid_line_mask(W, H, list(grid_shape)))
    out = np.zeros(((G // 2) * W, H, 3), np.uint8)
    lib().ora_ig_render(W, H, R, _p(disc), G, _p(sh), _p(cells), _p(tg), len(tc), _p(tc), _p(lines), _p(out))
    return out


def image_draw_philox(cfg, seed, env, tick, stream, n):
    """Philox streams: the transforms (R, cx, cy, angle, flip) of the n images drawn at one tick, in order."""
    out = np.zeros((n, 5), np.int32)
    lib().ora_i_draw_philox(C.byref(cfg), int(seed), int(env), int(tick), int(stream), int(n), _p(out))
    return [tuple(int(v) for v in row) for row in out]


def image_draw(cfg, rng_words):
    """Advance the image-space RNG by one observation; returns (R, cx, cy, angle, flip)."""
    vals = [C.c_int() for _ in range(5)]
    lib().ora_i_draw(C.byref(cfg), _p(rng_words), *[C.byref(v) for v in vals])
    return tuple(v.value for v in vals)


def image_rotate_flip_transpose(src, angle, flip):
    H, W = src.shape
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.zeros((W, H), np.uint8)
    lib().ora_i_rotate_flip_transpose(W, H, _p(src), int(angle), int(flip), _p(out))
    return out
