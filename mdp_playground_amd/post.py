"""VectorPostProcessor — the reference's GymEnvWrapper (mdp_playground/envs/gym_env_wrapper.py) as a
batched device-side post-processor for ANY batched simulator's (obs, reward, done) tensors.

Instance i of the batch is one reference object `GymEnvWrapper(env_i, **{**config, "seed": seed + i})`:
the same config keys (state_space_type, delay, transition_noise, reward_noise, reward_scale,
reward_shift, term_state_reward, image_transforms, image_padding, image_sh_quant, seed), the same
generator (seeded like the wrapper's __init__, :93, incl. the two space seeds it draws first, :102-103)
consumed in the same order, so outputs are bit-identical to the reference's under identical seeds.

    post = VectorPostProcessor(num_envs, n_actions=env.action_space.n, **config)
    obs = post.reset(first_obs)                       # images: the padded / shifted canvas
    a_env = post.actions(a)                           # discrete action noise (identity without it)
    obs, reward = post.step(base_obs, base_reward, base_done)

Differences from upstream, all documented in INTEGRATION.md: the `done` branch (:407-414) raises
TypeError upstream (list * float); implemented is its numpy meaning.  Callable noise functions and
Atari preprocessing are host-side composition, not here; the nested irrelevant-features toy env is
`IrrelevantToyEnvWrapper` below (two batched envs side by side on the device).
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch

from . import _capi as capi
from . import mdp as mdp_mod


class VectorPostProcessor:
    def __init__(self, num_envs, device=None, *, n_actions=0, obs_shape=None, obs_dtype=None, rng="numpy",
                 env_id_offset=0, philox_seed=None, autoreset=False, draws_before_use=2, **config):
        self._lib = capi.load()
        self._h = None
        if not torch.cuda.is_available():
            raise capi.MdppError("VectorPostProcessor needs a ROCm GPU; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_envs, self.config = int(num_envs), dict(config)
        sst = config.get("state_space_type")
        if sst not in ("discrete", "continuous"):
            raise ValueError("config['state_space_type'] must be 'discrete' or 'continuous'")     # :99, :354
        self.continuous = sst == "continuous"
        for key in ("transition_noise", "reward_noise"):
            if callable(config.get(key)):
                raise NotImplementedError(f"callable {key}: a Python function cannot run in a kernel")
        cfg = capi.MdppPostConfig()
        cfg.abi_version, cfg.num_envs, cfg.env_id_offset = capi.MDPP_ABI_VERSION, self.num_envs, int(env_id_offset)
        cfg.rng_mode = capi.RNG_NUMPY_PCG64 if rng == "numpy" else capi.RNG_PHILOX
        self.rng = rng
        seed = config.get("seed")
        cfg.philox_seed = (int(seed) if philox_seed is None and seed is not None else int(philox_seed or 0x9E3779B97F4A7C15)) & (2 ** 64 - 1)
        cfg.continuous = int(self.continuous)
        cfg.n_actions = int(n_actions)
        cfg.delay = int(config.get("delay", 0))
        assert cfg.delay >= 0                                                   # :91
        tn = config.get("transition_noise")
        cfg.has_transition_noise, cfg.transition_noise = int(tn is not None), float(tn or 0.0)
        if not self.continuous and tn is not None:
            assert 0.0 <= tn <= 1.0, "transition_noise must be a value in [0.0, 1.0] when env is discrete, it was:" + str(tn)  # :105
            if n_actions < 2:
                raise ValueError("discrete transition_noise needs n_actions (env.action_space.n) >= 2")
        rn = config.get("reward_noise")
        cfg.has_reward_noise, cfg.reward_noise = int(rn is not None), float(rn or 0.0)
        cfg.reward_scale = float(config.get("reward_scale", 1.0))
        cfg.reward_shift = float(config.get("reward_shift", 0.0))
        cfg.term_state_reward = float(config.get("term_state_reward", 0.0))
        cfg.autoreset = int(bool(autoreset))
        self.image = bool(config.get("image_transforms"))
        self._obs_out_shape = None
        if self.continuous:
            if obs_shape is None or len(obs_shape) != 1:
                raise ValueError("continuous: obs_shape=(D,) needed")
            dt = np.dtype(obs_dtype or np.float32)
            if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
                raise TypeError("continuous observations must be float32 or float64")
            cfg.obs_dim, cfg.obs_f64 = int(obs_shape[0]), int(dt == np.float64)
            self._obs_torch = torch.float64 if cfg.obs_f64 else torch.float32
            self._obs_in_shape = self._obs_out_shape = (cfg.obs_dim,)
        if self.image:
            assert not self.continuous, "Image transforms are only supported for discrete envs with image observations."  # :135
            if obs_shape is None or len(obs_shape) != 3:
                raise ValueError("image_transforms: obs_shape=(H, W, C) needed")
            H, W, Cc = (int(x) for x in obs_shape)
            assert H == W, "Currently only square images are supported."        # :531
            pad = int(config.get("image_padding", 20))                            # :148-151
            cfg.image, cfg.img_h, cfg.img_w, cfg.img_c, cfg.img_pad = 1, H, W, Cc, pad
            cfg.img_has_shift = int("shift" in config["image_transforms"])
            cfg.img_sh_quant = int(config.get("image_sh_quant") or 1)            # :153-163
            self._obs_torch = torch.uint8
            self._obs_in_shape, self._obs_out_shape = (H, W, Cc), (W + 2 * pad, H + 2 * pad, Cc)
        h = C.c_void_p()
        rc = self._lib.mdpp_post_create(C.byref(cfg), self.device.index, C.byref(h))
        if rc:
            msg = self._lib.mdpp_post_last_error(None)
            raise capi.MdppError(f"mdpp_post_create failed ({rc}): {msg.decode() if msg else ''}")
        self._h, self._cfg = h, cfg
        self._draws_before_use = int(draws_before_use)
        if rng == "numpy":
            self.seed(seed)

    # ------------------------------------------------------------------
    def _check(self, rc, what):
        if rc:
            msg = self._lib.mdpp_post_last_error(self._h)
            raise capi.MdppError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def seed(self, seed=None):
        """seed(seed) (:488-509) as __init__ uses it (:93-104): instance i's generator is
        PCG64(SeedSequence(seed + i)) advanced by the draws the wrapper's constructor makes before any
        step -- the observation- and action-space seeds, `integers(sys.maxsize)` each (:102-103; three with
        an irrelevant toy env, :234: pass draws_before_use=3)."""
        if self.rng != "numpy":
            raise capi.MdppError("seed() re-seeds numpy streams; this post-processor uses rng='philox'")
        if seed is None:
            seed = int(np.random.SeedSequence().entropy)
        words = np.zeros((self.num_envs, 6), np.uint64)
        off = int(self._cfg.env_id_offset)
        for i in range(self.num_envs):
            g = mdp_mod.new_generator(int(seed) + off + i)
            for _ in range(self._draws_before_use):
                g.integers(sys.maxsize)
            words[i] = mdp_mod.pcg64_words(g)
        self.put_streams(words)
        return seed

    def put_streams(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape == (self.num_envs, 6)
        self._check(self._lib.mdpp_post_seed_streams(self._h, capi.nptr(words)), "mdpp_post_seed_streams")

    def get_streams(self):
        words = np.zeros((self.num_envs, 6), np.uint64)
        self._check(self._lib.mdpp_post_get_streams(self._h, capi.nptr(words)), "mdpp_post_get_streams")
        return words

    def reward_buffer(self):
        ring = np.zeros((self.num_envs, int(self._cfg.delay)), np.float64)
        self._check(self._lib.mdpp_post_get_reward_buffer(self._h, capi.nptr(ring)), "mdpp_post_get_reward_buffer")
        return ring

    def _obs_args(self, obs, lead, out):
        if self._obs_out_shape is None:
            return None, None, obs
        want = tuple(lead) + tuple(self._obs_in_shape)
        if not torch.is_tensor(obs):
            obs = torch.as_tensor(np.asarray(obs), device=self.device)
        if obs.dtype != self._obs_torch or tuple(obs.shape) != want:
            raise ValueError(f"observations must be {self._obs_torch} of shape {want}, got {obs.dtype} {tuple(obs.shape)}")
        obs = obs.to(self.device).contiguous()
        if out is None:
            out = torch.empty(tuple(lead) + tuple(self._obs_out_shape), dtype=self._obs_torch, device=self.device)
        return C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr()), out

    def reset(self, obs=None, mask=None, out=None):
        """reset() (:441-486): the reward buffer refilled with zeros; image handles return the canvas of the
        first observation (instances outside `mask` keep what `out` held)."""
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        if self.image:
            pin, pout, out = self._obs_args(obs, (self.num_envs,), out)
        else:
            pin = pout = None
            out = obs
        self._check(self._lib.mdpp_post_reset(self._h, mptr, pin, pout, self._stream()), "mdpp_post_reset")
        return out

    def actions(self, actions, out=None):
        """The actions the inner envs receive (:354-366); continuous actions pass through."""
        if self.continuous:
            return actions
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if tuple(a.shape) != (self.num_envs,):
            raise ValueError(f"actions must have shape ({self.num_envs},)")
        out = torch.empty_like(a) if out is None else out
        self._check(self._lib.mdpp_post_actions(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()),
                    "mdpp_post_actions")
        return out

    def step(self, obs, reward, done, out=None):
        """One step of post-processing (:367-432) -> (obs, reward float64[N]).  Accepts a leading K axis
        ([K, N, ...], time-major) for K fused steps."""
        r = torch.as_tensor(reward, device=self.device).to(torch.float64).contiguous()
        d = torch.as_tensor(done, device=self.device).to(torch.uint8).contiguous()
        if r.dim() == 1:
            lead, K = (self.num_envs,), 1
        else:
            lead, K = (int(r.shape[0]), self.num_envs), int(r.shape[0])
        if tuple(r.shape) != lead or tuple(d.shape) != lead:
            raise ValueError(f"reward and done must have shape {lead}")
        pin, pout, obs_out = self._obs_args(obs, lead, out)
        r_out = torch.empty_like(r)
        self._check(self._lib.mdpp_post_step_n(self._h, K, pin, C.c_void_p(r.data_ptr()), C.c_void_p(d.data_ptr()), pout,
                                               C.c_void_p(r_out.data_ptr()), self._stream()), "mdpp_post_step_n")
        return obs_out, r_out

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize(self.device)
            self._lib.mdpp_post_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IrrelevantToyEnvWrapper:
    """The wrapper's nested irrelevant toy env (gym_env_wrapper.py:214-270, :378-396, :476-486): config key
    "irrelevant_features" holds an RLToyEnv config; its env steps beside the wrapped env on the second part of every
    action and contributes the second part of every observation -- nothing else (its reward and done are ignored, it is
    only ever reset together with the wrapped env).  Batched: `env` is any batched env on the device with
    reset(seed=None, mask=None) -> (obs, info) and step(actions) -> (obs, reward, terminated, truncated, info) -- e.g. an
    RLToyVectorEnv with autoreset="disabled" -- and the toy env an `RLToyVectorEnv(num_envs, **irrelevant_features)`.

    discrete (:379-383): actions int32 [N, 2] = (wrapped, toy); observations [N, 2]
    continuous (:384-396): actions float32 [N, Da + Db], the first Da (`env_action_dim`) for the wrapped env;
        observations [N, Do + Db], concatenated.
    autoreset=True: instances whose wrapped env reports terminated | truncated are reset (both envs) in the same call,
    the returned observation is the new episode's first (info["final_obs"]: the terminal one)."""

    def __init__(self, env, num_envs, irrelevant_features, *, state_space_type, env_action_dim=None, autoreset=False,
                 device=None, **toy_kwargs):
        from .vector_env import RLToyVectorEnv
        if state_space_type not in ("discrete", "continuous"):
            raise ValueError("state_space_type must be 'discrete' or 'continuous'")
        self.env, self.num_envs, self.continuous = env, int(num_envs), state_space_type == "continuous"
        self.autoreset = bool(autoreset)
        toy_kwargs.setdefault("autoreset", "disabled")            # it keeps stepping after its own `done` (:380, :391)
        self.irr_toy_env = RLToyVectorEnv(num_envs=self.num_envs, device=device, **toy_kwargs, **irrelevant_features)
        if self.continuous and env_action_dim is None:
            raise ValueError("continuous: env_action_dim (action dimensions of the wrapped env) is needed")
        self.env_action_dim = env_action_dim

    def _join(self, o, o_irr):
        if self.continuous:
            return torch.cat((o, o_irr.to(o.dtype)), dim=1)
        return torch.stack((o.to(torch.int64), o_irr.to(torch.int64)), dim=1)

    def reset(self, seed=None, mask=None):
        kw = {} if mask is None else {"mask": mask}
        o, info = self.env.reset(seed=seed, **kw)
        o_irr, info_irr = self.irr_toy_env.reset(seed=seed, **kw)
        return self._join(o, o_irr), (info, info_irr)

    def step(self, actions):
        if self.continuous:
            a0, a1 = actions[:, :self.env_action_dim].contiguous(), actions[:, self.env_action_dim:].contiguous()
        else:
            a0, a1 = actions[:, 0].contiguous(), actions[:, 1].contiguous()
        o, r, term, trunc, info = self.env.step(a0)
        o_irr, _, _, _, _ = self.irr_toy_env.step(a1)
        obs = self._join(o, o_irr)
        if self.autoreset:
            ended = term | trunc
            info = dict(info or {}, final_obs=obs)
            if bool(ended.any()):
                fresh, _ = self.reset(mask=ended)
                obs = torch.where(ended.view(-1, *([1] * (obs.dim() - 1))), fresh, obs)
        return obs, r, term, trunc, info

    def close(self):
        self.irr_toy_env.close()
        if hasattr(self.env, "close"):
            self.env.close()
