"""RLToyVectorEnv — the reference's RLToyEnv (mdp_playground/envs/rl_toy_env.py:26) batched on
one MI355X: same config dict, Gym-style reset/step, torch tensors in and out.

Each env instance i of the batch behaves exactly like one reference object:

* ``RLToyVectorEnv(num_envs=N, **config)`` — one MDP (tables generated on the host from
  ``config["seed"]`` exactly as ``RLToyEnv.__init__`` does) shared by N instances.  Instance i is
  the reference env built from ``config`` and then ``reset(seed=seed_dict["env"] + i)`` (what a
  gymnasium SyncVectorEnv does); its space generator (discrete P-noise, continuous reset
  sampling) is the reference's own for i = 0 and ``Space.seed(space_seed + i)`` for i > 0.
* ``RLToyVectorEnv(seeds=[s0, s1, ...], **config)`` — N *different* MDPs: instance i is the
  reference ``RLToyEnv(**{**config, "seed": seeds[i]})`` verbatim (tables per env in HBM).

``rng="numpy"`` (default) keeps numpy ``Generator(PCG64)`` streams per env on the device, so
noise and resets are bit-identical to the reference under identical seeds; ``rng="philox"`` is a
stateless counter-based alternative (no RNG state in HBM, own stream, same distributions).
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np
import torch

from . import _capi as capi
from . import mdp as mdp_mod
from .spaces import BatchedSpace, BoxSpace, DiscreteSpace, ImageSpace, TupleSpace

_AUTORESET = {"disabled": capi.AUTORESET_DISABLED, "same_step": capi.AUTORESET_SAME_STEP,
              "next_step": capi.AUTORESET_NEXT_STEP}


def _stack(arrs, dtype):
    return np.ascontiguousarray(np.stack([np.asarray(a) for a in arrs]), dtype=dtype)


class RLToyVectorEnv:
    metadata = {"render_modes": []}

    def __init__(self, num_envs=None, device=None, *, seeds=None, rng="numpy",
                 autoreset="same_step", max_episode_steps=None, env_id_offset=0,
                 philox_seed=None, episode_stats=False, mdps=None, **config):
        self._lib = capi.load()
        self._h = None
        if not torch.cuda.is_available():
            raise capi.MdppError("RLToyVectorEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        if autoreset not in _AUTORESET:
            raise ValueError("autoreset must be 'same_step', 'next_step' or 'disabled'")
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise capi.MdppError("device must be a cuda (ROCm) device")
        if self.device.index is None:          # 'cuda' -> 'cuda:<current>': step() compares tensors' devices with it
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.config = copy.deepcopy(config)   # the reference mutates its config (:339,...); we do not
        if seeds is not None:
            if num_envs is not None and num_envs != len(seeds):
                raise ValueError("num_envs != len(seeds)")
            num_envs = len(seeds)
            if mdps is not None:          # prebuilt (mdp.build_many, before the GPU was touched): one per seed, in order
                if len(mdps) != len(seeds):
                    raise ValueError("len(mdps) != len(seeds)")
                self.mdps = list(mdps)
            else:
                self.mdps = [mdp_mod.build_mdp({**config, "seed": s}) for s in seeds]
        else:
            if num_envs is None:
                num_envs = 1
            self.mdps = [mdp_mod.build_mdp(config)]
        self.num_envs = int(num_envs)
        self.env_id_offset = int(env_id_offset)
        self.autoreset = autoreset
        self.rng = rng
        m = self.mdps[0]
        self.kind = m.kind
        self.seed_dict = m.seed_dict
        self._per_env = seeds is not None
        self.seeded_streams = {}

        cfg = capi.MdppConfig()
        cfg.abi_version = capi.MDPP_ABI_VERSION
        cfg.kind = {"discrete": capi.KIND_DISCRETE, "continuous": capi.KIND_CONTINUOUS,
                    "grid": capi.KIND_GRID}[m.kind]
        cfg.num_envs = self.num_envs
        cfg.env_id_offset = self.env_id_offset
        cfg.rng_mode = capi.RNG_NUMPY_PCG64 if rng == "numpy" else capi.RNG_PHILOX
        cfg.autoreset = _AUTORESET[autoreset]
        cfg.max_episode_steps = int(max_episode_steps or 0)
        if philox_seed is None:
            e = m.seed_dict.get("env")
            philox_seed = 0x9E3779B97F4A7C15 if e is None else int(e)
        cfg.philox_seed = int(philox_seed) & (2 ** 64 - 1)
        cfg.delay = int(m.delay)
        cfg.every_n = int(m.reward_every_n_steps)
        cfg.has_reward_noise = int(m.reward_noise is not None)
        cfg.reward_noise = float(m.reward_noise or 0.0)
        cfg.reward_scale = float(m.reward_scale)
        cfg.reward_shift = float(m.reward_shift)
        cfg.term_state_reward = float(m.term_state_reward)
        # the reference's per-episode noise statistics (logged at every reset(), rl_toy_env.py:2231-2247), per env;
        # such handles run on the general kernels
        self.episode_stats = bool(episode_stats)
        cfg.episode_stats = int(self.episode_stats)
        self._irr = False
        if m.kind == "discrete":
            self._init_discrete(cfg)
        elif m.kind == "grid":
            self._init_grid(cfg)
        else:
            self._init_continuous(cfg)
        h = C.c_void_p()
        rc = self._lib.mdpp_create(C.byref(cfg), self.device.index, C.byref(h))
        capi.check(self._lib, None, rc, "mdpp_create")
        self._h = h
        self._cfg = cfg
        if m.kind == "discrete":
            self._upload_discrete()
        elif self._image is not None:
            disc = np.ascontiguousarray(self._image["disc"], dtype=np.uint8)
            rc = self._lib.mdpp_upload_image_disc(self._h, capi.nptr(disc))
            capi.check(self._lib, self._h, rc, "mdpp_upload_image_disc")
            if m.kind == "grid":
                lines = np.ascontiguousarray(self._image["lines"], dtype=np.uint8)
                rc = self._lib.mdpp_upload_image_lines(self._h, capi.nptr(lines))
                capi.check(self._lib, self._h, rc, "mdpp_upload_image_lines")
        self._alloc_buffers()
        if rng == "numpy":
            self._seed_streams(self.seed_dict.get("env"), initial=True)
        # the reference constructor ends with reset(seed=seed_dict["env"]) (:831-833)
        self._reset_all()

    # ------------------------------------------------------------------ construction helpers
    def _init_discrete(self, cfg):
        m = self.mdps[0]
        for o in self.mdps[1:]:
            if (o.S, o.A, o.sequence_length, o.delay, o.S_irr, o.A_irr) != \
                    (m.S, m.A, m.sequence_length, m.delay, m.S_irr, m.A_irr):
                raise ValueError("per-env MDPs must share S, A, sequence_length and delay")
        cfg.S, cfg.A, cfg.L = m.S, m.A, m.sequence_length
        cfg.num_tables = self.num_envs if self._per_env else 1
        unit = all(v == 1.0 for mm in self.mdps for k, v in mm.rewardable_sequences.items()
                   if len(k) == mm.sequence_length)
        custom_r = [mm.reward_matrix is not None for mm in self.mdps]
        if any(custom_r) and not all(custom_r):
            raise ValueError("per-env MDPs must all use a reward matrix or all use rewardable sequences")
        cfg.reward_kind = capi.REWARD_STATE_ACTION if custom_r[0] else capi.REWARD_SEQUENCES
        cfg.unit_rewards = int(unit and m.delay <= 32 and not custom_r[0])
        cfg.has_transition_noise = int(bool(m.transition_noise))
        cfg.transition_noise = float(m.transition_noise or 0.0)
        dtype_o = self.config.get("dtype_o", self.config.get("dtype_s", np.int64))
        self._obs_torch_dtype = torch.int32 if np.dtype(dtype_o) == np.int32 else torch.int64
        cfg.obs_dtype = capi.OBS_I32 if self._obs_torch_dtype == torch.int32 else capi.OBS_I64
        self.single_observation_space = DiscreteSpace(m.S, seed=m.seed_dict.get("relevant_state_space"))
        self.single_action_space = DiscreteSpace(m.A, seed=m.seed_dict.get("relevant_action_space"))
        self._irr = bool(m.irrelevant)
        if self._irr:
            # Tuple spaces (:718-733): (relevant, irrelevant) pairs
            cfg.irrelevant, cfg.S_irr, cfg.A_irr = 1, m.S_irr, m.A_irr
            self.single_observation_space = TupleSpace(
                [self.single_observation_space,
                 DiscreteSpace(m.S_irr, seed=m.seed_dict.get("irrelevant_state_space"))],
                seed=m.seed_dict.get("state_space"))
            self.single_action_space = TupleSpace(
                [self.single_action_space,
                 DiscreteSpace(m.A_irr, seed=m.seed_dict.get("irrelevant_action_space"))],
                seed=m.seed_dict.get("action_space"))
        self._image = None
        if m.image is not None:
            from . import image_obs
            im = m.image
            # (with an irrelevant sub-space the observation is one image per sub-space, side by side
            # along x; state s is an (s + 3)-gon in either, so the templates cover the larger one)
            self._image = image_obs.build_templates(max(m.S, m.S_irr) if self._irr else m.S, im)
            cfg.image, cfg.img_w, cfg.img_h = 1, im["width"], im["height"]
            tr = im["transforms"]
            cfg.img_has_scale, cfg.img_has_shift = int("scale" in tr), int("shift" in tr)
            cfg.img_has_rotate, cfg.img_has_flip = int("rotate" in tr), int("flip" in tr)
            cfg.img_sh_quant = int(im["sh_quant"] or 1)
            cfg.img_ro_quant = int(im["ro_quant"] or 1)
            cfg.img_r0 = im["circle_radius"]
            cfg.img_r_min, cfg.img_r_max = self._image["r_min"], self._image["r_max"]
            cfg.img_log_min_r, cfg.img_log_max_r = self._image["log_min_r"], self._image["log_max_r"]
            cfg.img_tpl_size = self._image["tpl_size"]
            cfg.obs_dtype = capi.OBS_IMAGE_U8
            self._obs_torch_dtype = torch.uint8
            self.single_observation_space = ImageSpace((2 if self._irr else 1) * im["width"], im["height"])
        self.transition_matrix = m.P
        self.rewardable_sequences = m.rewardable_sequences
        self.reward_matrix = m.reward_matrix               # use_custom_mdp with matrices (:1259-1267)

    def _upload_discrete(self):
        ms = self.mdps
        unit = bool(self._cfg.unit_rewards)
        P = _stack([m.P for m in ms], np.uint8 if ms[0].S <= 255 else np.uint16)      # (S > 255: 16-bit entries, mdpp_discrete_wide.hip)
        is_term = _stack([m.is_terminal_table() for m in ms], np.uint8)
        init_cdf = _stack([m.init_cdf() for m in ms], np.float64)
        rtable = rbits = None
        if unit:
            rbits = _stack([np.packbits((m.reward_table() != 0).astype(np.uint8), bitorder="little")
                            for m in ms], np.uint8)
        else:
            rtable = _stack([m.reward_table() for m in ms], np.float64)
        noise = ms[0].noise_cdf()
        noise = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
        rc = self._lib.mdpp_upload_discrete_tables(self._h, capi.nptr(P), capi.nptr(rtable),
                                                   capi.nptr(rbits), capi.nptr(is_term),
                                                   capi.nptr(init_cdf), capi.nptr(noise))
        capi.check(self._lib, self._h, rc, "mdpp_upload_discrete_tables")
        if self._irr:
            P1 = _stack([m.P_irr for m in ms], np.uint8)
            cdf1 = _stack([m.init_cdf_irr() for m in ms], np.float64)
            noise1 = ms[0].noise_cdf(irrelevant=True)
            noise1 = None if noise1 is None else np.ascontiguousarray(noise1, dtype=np.float64)
            rc = self._lib.mdpp_upload_discrete_irrelevant(self._h, capi.nptr(P1), capi.nptr(cdf1),
                                                           capi.nptr(noise1))
            capi.check(self._lib, self._h, rc, "mdpp_upload_discrete_irrelevant")
        if self._image is not None:
            t = self._image
            tpl = np.ascontiguousarray(t["tpl"], dtype=np.uint8)
            cx = np.ascontiguousarray(t["cls_x"], dtype=np.int16)
            cy = np.ascontiguousarray(t["cls_y"], dtype=np.int16)
            rc = self._lib.mdpp_upload_image_templates(self._h, capi.nptr(tpl), tpl.shape[1],
                                                       t["n_cls_x"], t["n_cls_y"], capi.nptr(cx),
                                                       capi.nptr(cy))
            capi.check(self._lib, self._h, rc, "mdpp_upload_image_templates")

    def _init_grid(self, cfg):
        """Grid envs (rl_toy_env.py:539-541, :780-811): int64 cell vectors, one-hot +-1 actions."""
        m = self.mdps[0]
        G = len(m.grid_shape)
        cfg.grid_dims = G
        for d in range(G):
            cfg.grid_shape[d] = int(m.grid_shape[d])
        cfg.grid_target[0], cfg.grid_target[1] = int(m.target_point[0]), int(m.target_point[1])
        cfg.make_denser = int(bool(m.make_denser))
        cfg.has_transition_noise = int(bool(m.transition_noise))
        cfg.transition_noise = float(m.transition_noise or 0.0)
        dtype_o = self.config.get("dtype_o", self.config.get("dtype_s", np.int64))
        self._obs_torch_dtype = torch.int32 if np.dtype(dtype_o) == np.int32 else torch.int64
        cfg.obs_dtype = capi.OBS_I32 if self._obs_torch_dtype == torch.int32 else capi.OBS_I64
        # Box(0, grid_shape, int64) / GridActionSpace(-1, 1) (:780-800)
        self.single_observation_space = BoxSpace(0, 1, (G,), dtype=np.int64, seed=m.seed_dict.get("state_space"))
        self.single_observation_space.high = np.asarray(m.grid_shape, dtype=np.int64)
        self.single_action_space = BoxSpace(-1, 1, (G,), dtype=np.int64, seed=m.seed_dict.get("action_space"))
        self._image = None
        if m.image is not None:
            # ImageContinuous with grid lines (:800-811): uint8 [n_sub * W][H][3]
            from . import image_obs
            im = m.image
            self._image = dict(im, disc=image_obs.disc_template(im["circle_radius"]), n_sub=G // 2,
                               lines=image_obs.grid_line_mask(im["width"], im["height"], list(m.grid_shape)))
            cfg.image, cfg.img_w, cfg.img_h, cfg.img_r0 = 1, im["width"], im["height"], im["circle_radius"]
            cells = m.terminal_states or []
            cfg.n_boxes = len(cells)
            for b, c in enumerate(cells):            # terminal cells (drawn, never terminating) ride in box_lo
                cfg.box_lo[2 * b], cfg.box_lo[2 * b + 1] = float(c[0]), float(c[1])
            cfg.obs_dtype = capi.OBS_IMAGE_U8
            self._obs_torch_dtype = torch.uint8
            self.single_observation_space = BoxSpace(0, 255, (self._image["n_sub"] * im["width"], im["height"], 3),
                                                     dtype=np.uint8)

    def _init_continuous(self, cfg):
        m = self.mdps[0]
        if self._per_env:
            # continuous MDPs carry no generated tables: per-env seeds only change the streams
            pass
        cfg.D, cfg.n_rel, cfg.order = m.D, len(m.relevant_indices), m.order
        self._line_L = 0
        if m.reward_function == "move_along_a_line":
            cfg.reward_function, cfg.L = capi.CREWARD_MOVE_ALONG_A_LINE, m.sequence_length
            self._line_L = int(m.sequence_length)
        for j, r in enumerate(m.relevant_indices):
            cfg.rel_idx[j] = int(r)
            cfg.target[j] = float(m.target_point[j])
        cfg.make_denser = int(bool(m.make_denser))
        cfg.target_f64 = int(bool(m.target_default))      # the default target_point: float64 zeros (:652-654)
        cfg.has_p_noise = int(m.transition_noise is not None)
        cfg.p_noise = float(m.transition_noise or 0.0)
        cfg.inertia, cfg.time_unit = float(m.inertia), float(m.time_unit)
        cfg.state_space_max, cfg.action_space_max = float(m.state_space_max), float(m.action_space_max)
        cfg.target_radius, cfg.action_loss_weight = float(m.target_radius), float(m.action_loss_weight)
        nb = 0 if m.box_lo is None else len(m.box_lo)
        # (the reference has no limit, rl_toy_env.py:891-956; here the cubes ride in the handle's argument block, [nb][n_rel]
        #  floats in arrays of MAX_BOXES * MAX_DIM: 64 cubes at four relevant dimensions; pictures draw them from a device list)
        if nb * cfg.n_rel > capi.MAX_BOXES * capi.MAX_DIM:
            raise NotImplementedError(f"at most {capi.MAX_BOXES * capi.MAX_DIM // max(cfg.n_rel, 1)} terminal hypercubes at "
                                      f"{cfg.n_rel} relevant dimensions")
        cfg.n_boxes = nb
        for b in range(nb):
            for j in range(cfg.n_rel):
                cfg.box_lo[b * cfg.n_rel + j] = float(m.box_lo[b][j])
                cfg.box_hi[b * cfg.n_rel + j] = float(m.box_hi[b][j])
        cfg.obs_dtype = capi.OBS_F32
        self._obs_torch_dtype = torch.float32
        self.single_observation_space = BoxSpace(-m.state_space_max, m.state_space_max, (m.D,),
                                                 seed=m.seed_dict.get("state_space"))
        self._image = None
        if m.image is not None:
            # ImageContinuous observations (:770-778): uint8 [n_sub * W][H][3]; no random transforms
            from . import image_obs
            im = m.image
            if not np.isfinite(m.state_space_max):
                raise AssertionError("ImageContinuous needs a bounded feature space")   # image_continuous.py:62-63
            self._image = dict(im, disc=image_obs.disc_template(im["circle_radius"]), n_sub=2 if m.D > 2 else 1)
            cfg.image, cfg.img_w, cfg.img_h, cfg.img_r0 = 1, im["width"], im["height"], im["circle_radius"]
            cfg.obs_dtype = capi.OBS_IMAGE_U8
            self._obs_torch_dtype = torch.uint8
            self.single_observation_space = BoxSpace(0, 255, (self._image["n_sub"] * im["width"], im["height"], 3),
                                                     dtype=np.uint8)
        self.single_action_space = BoxSpace(-m.action_space_max, m.action_space_max, (m.D,),
                                            seed=m.seed_dict.get("action_space"))

    def _alloc_buffers(self):
        N, dev = self.num_envs, self.device
        shape = self._obs_shape(N)
        self._obs = torch.zeros(shape, dtype=self._obs_torch_dtype, device=dev)
        self._final_obs = torch.zeros(shape, dtype=self._obs_torch_dtype, device=dev)
        self._reward = torch.zeros(N, dtype=torch.float32, device=dev)
        self._term = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._trunc = torch.zeros(N, dtype=torch.uint8, device=dev)
        # batched spaces, as gymnasium's VectorEnv exposes them next to the single_* ones
        self.observation_space = BatchedSpace(self.single_observation_space, N)
        self.action_space = BatchedSpace(self.single_action_space, N)
        # everything step() passes on every call, prepared once (the call itself is the hot path)
        self._mdpp_step = self._lib.mdpp_step
        self._p_obs, self._p_final = self._obs.data_ptr(), self._final_obs.data_ptr()
        self._p_reward, self._p_term, self._p_trunc = (self._reward.data_ptr(), self._term.data_ptr(),
                                                        self._trunc.data_ptr())
        self._term_b, self._trunc_b = self._term.view(torch.bool), self._trunc.view(torch.bool)
        self._info = {"final_obs": self._final_obs} if self.autoreset == "same_step" else {}
        self._act_dtype = torch.float32 if self.kind == "continuous" else torch.int32
        if self.kind == "discrete":
            self._act_shape = torch.Size((N, 2) if self._irr else (N,))
        elif self.kind == "grid":
            self._act_shape = torch.Size((N, len(self.mdps[0].grid_shape)))
        else:
            self._act_shape = torch.Size((N, self.mdps[0].D))

    def _obs_shape(self, *lead):
        if self.kind == "continuous":
            if getattr(self, "_image", None) is not None:
                im = self._image
                return tuple(lead) + (im["n_sub"] * im["width"], im["height"], 3)
            return tuple(lead) + (self.mdps[0].D,)
        if self.kind == "grid":
            if getattr(self, "_image", None) is not None:
                im = self._image
                return tuple(lead) + (im["n_sub"] * im["width"], im["height"], 3)
            return tuple(lead) + (len(self.mdps[0].grid_shape),)
        if getattr(self, "_image", None) is not None:
            im = self.mdps[0].image
            return tuple(lead) + ((2 if getattr(self, "_irr", False) else 1) * im["width"], im["height"], 1)
        if getattr(self, "_irr", False):
            return tuple(lead) + (2,)
        return tuple(lead)

    def _seed_streams(self, env_seed, initial):
        """Env streams: PCG64(SeedSequence(env_seed + global id)), i.e. what reset(seed=...) does
        (gymnasium Env.reset -> seeding.np_random; rl_toy_env.py:2225).  Space streams are only
        (re)built at construction."""
        N, off = self.num_envs, self.env_id_offset
        if self._per_env:
            env_words = np.stack([mdp_mod.pcg64_words(mdp_mod.new_generator(
                m.seed_dict["env"] if env_seed is None or initial else env_seed + off + i))
                for i, m in enumerate(self.mdps)])
        else:
            base = None if env_seed is None else env_seed + off
            env_words = mdp_mod.fresh_stream_words(base, N)
        self._put_stream(capi.STREAM_ENV, env_words)
        if not initial:
            return
        if self.kind == "discrete":
            if self._per_env:
                sp = np.stack([m.space_rng_words for m in self.mdps])
            else:
                # (with an irrelevant sub-space the Tuple re-seeded both state spaces: space_seeds)
                R = self.mdps[0].space_seeds[0]
                sp = mdp_mod.fresh_stream_words(R + off, N)
                if off == 0:
                    sp[0] = self.mdps[0].space_rng_words   # the reference's own generator, post-P
        else:
            if self._per_env:
                sp = np.stack([mdp_mod.pcg64_words(mdp_mod.new_generator(m.seed_dict["state_space"]))
                               for m in self.mdps])
            else:
                sp = mdp_mod.fresh_stream_words(self.seed_dict["state_space"] + off, N)
        self._put_stream(capi.STREAM_SPACE, sp)
        if self.kind == "grid":                      # GridActionSpace(seed=seed_dict["action_space"]), :793-797
            if self._per_env:
                ac = np.stack([mdp_mod.pcg64_words(mdp_mod.new_generator(m.seed_dict["action_space"]))
                               for m in self.mdps])
            else:
                ac = mdp_mod.fresh_stream_words(self.seed_dict["action_space"] + off, N)
            self._put_stream(capi.STREAM_ACTION, ac)
        if self.kind == "discrete" and self._irr:
            if self._per_env:
                sp1 = np.stack([m.space_irr_rng_words for m in self.mdps])
            else:
                sp1 = mdp_mod.fresh_stream_words(self.mdps[0].space_seeds[1] + off, N)
                if off == 0:
                    sp1[0] = self.mdps[0].space_irr_rng_words
            self._put_stream(capi.STREAM_SPACE_IRR, sp1)
        if self.kind == "discrete" and self._image is not None:
            if self._per_env:
                im = np.stack([mdp_mod.pcg64_words(mdp_mod.new_generator(m.image["seed"])) for m in self.mdps])
            else:
                im = mdp_mod.fresh_stream_words(self.mdps[0].image["seed"] + off, N)
            self._put_stream(capi.STREAM_IMAGE, im)

    def _put_stream(self, stream, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape == (self.num_envs, 6)
        self.seeded_streams[stream] = words.copy()   # what was last uploaded (tests, checkpoints)
        rc = self._lib.mdpp_seed_streams(self._h, stream, capi.nptr(words))
        capi.check(self._lib, self._h, rc, "mdpp_seed_streams")

    # ------------------------------------------------------------------ Gym-style API
    def _stream(self):
        return C.c_void_p(self._raw_stream())

    def _raw_stream(self):
        """The caller's current HIP stream as an integer.  torch.cuda.current_stream() builds a Stream object (1.8 us per
        call on the GPU box, tools/host_cost.py) -- a third of a single step's host time; the raw getter takes 0.1 us."""
        get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if get is not None:
            return get(self.device.index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _reset_all(self, mask=None):
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
            src = getattr(self, "_obs_src", None)
            if src is not None:          # the rows of the envs NOT reset: their observation after the last rollout() / graph replay
                self._obs.copy_(src)
        self._obs_src = None
        rc = self._lib.mdpp_reset(self._h, mptr, C.c_void_p(self._obs.data_ptr()), self._stream())
        capi.check(self._lib, self._h, rc, "mdpp_reset")
        return self._obs

    def reset(self, seed=None, options=None, mask=None):
        """reset(seed=None) -> (obs, {}), rl_toy_env.py:2217.  ``seed`` re-seeds env i's generator
        with seed + i first, as the reference (gymnasium Env.reset) does.  ``mask`` (bool[N])
        restricts the reset to some envs."""
        if seed is not None:
            if self.rng != "numpy":
                raise capi.MdppError("reset(seed=...) re-seeds numpy streams; this env uses rng='philox'")
            if not isinstance(seed, int) or seed < 0:
                raise TypeError("seed must be a non-negative python int")
            self._seed_streams(seed, initial=False)
        return self._reset_all(mask), {}

    def seed(self, seed=None):
        """seed(seed) -> int, rl_toy_env.py:2379-2406: re-seeds the env generators (not the spaces') —
        env i of the batch gets seed + i, like reset(seed=...) — and returns the seed; None draws one
        from OS entropy as gymnasium's seeding.np_random does."""
        if self.rng != "numpy":
            raise capi.MdppError("seed() re-seeds numpy streams; this env uses rng='philox'")
        if seed is None:
            seed = int(np.random.SeedSequence().entropy)
        if not isinstance(seed, int) or seed < 0:
            raise TypeError("seed must be a non-negative python int")    # gymnasium.utils.seeding.np_random
        self._seed_streams(seed, initial=False)
        self.seed_ = seed
        return seed

    def step_graph(self, actions):
        """A replayable HIP graph of K single steps (K = actions.shape[0] mdpp_step launches captured
        once): the one-launch-per-step API without the per-call host cost.  actions: [K, N, ...] as for
        rollout(); the tensor is read at every replay, so writing new actions into it between replays
        steps with them.  Returns a StepGraph with .replay() and the output buffers
        .obs/.reward/.terminated/.truncated ([K, N, ...]).

        The handle's step counter travels BY VALUE into captured launches (include/mdpp.h).  Where that would not replay
        exactly -- Philox streams key their draws by the counter, a delay line kept in memory starts at counter mod delay --
        the launches are captured in the library's capture mode instead: they add a device word to the counter, which
        replay() sets to (counter now - counter at capture) right before the graph (mdpp_graph_capture /
        mdpp_graph_set_tick_offset, round 4; image handles too since round 5)."""
        K = int(actions.shape[0])
        ok = self._lib.mdpp_graph_replay_exact(self._h, K)
        if ok < 0:
            capi.check(self._lib, self._h, ok, "mdpp_graph_replay_exact")
        if ok == 0:
            raise capi.MdppError(
                "step_graph: a captured graph of %d steps does not replay exactly for this handle (delay = %d). "
                "Use rollout() (one fused launch) instead." % (K, self._cfg.delay))
        by_offset = ok == 2
        a = self._as_actions(actions, K)
        obs, rew, term, trunc = self.alloc_rollout(K)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        g = torch.cuda.CUDAGraph()
        step, h = self._lib.mdpp_step, self._h

        def launches(stream):
            for k in range(K):
                rc = step(h, a[k].data_ptr(), obs[k].data_ptr(), rew[k].data_ptr(), term[k].data_ptr(),
                          trunc[k].data_ptr(), None, stream)
                if rc:
                    capi.check(self._lib, h, rc, "mdpp_step")
        tick0 = C.c_uint64()
        capi.check(self._lib, h, self._lib.mdpp_tick(h, 0, C.byref(tick0)), "mdpp_tick")
        try:
            if by_offset:
                capi.check(self._lib, h, self._lib.mdpp_graph_capture(h, 1), "mdpp_graph_capture")
            with torch.cuda.graph(g, stream=side):
                launches(side.cuda_stream)
        finally:
            if by_offset:
                self._lib.mdpp_graph_capture(h, 0)
            # the capture advanced the counter although nothing ran
            now = C.c_uint64()
            self._lib.mdpp_tick(h, 0, C.byref(now))
            self._lib.mdpp_tick(h, int(tick0.value) - int(now.value), None)
        return StepGraph(self, g, K, int(tick0.value), a, obs, rew, term, trunc, by_offset)

    def step(self, actions):
        """step(actions) -> (obs, reward, terminated, truncated, info), rl_toy_env.py:1992.
        Returned tensors alias preallocated device buffers (valid until the next call)."""
        a = actions
        if not (torch.is_tensor(a) and a.dtype == self._act_dtype and a.device == self.device
                and a.shape == self._act_shape and a.is_contiguous()):
            a = self._as_actions(actions, None)
        rc = self._mdpp_step(self._h, a.data_ptr(), self._p_obs, self._p_reward, self._p_term,
                             self._p_trunc, self._p_final, self._raw_stream())
        if rc:
            capi.check(self._lib, self._h, rc, "mdpp_step")
        self._obs_src = None
        return self._obs, self._reward, self._term_b, self._trunc_b, self._info

    def rollout(self, actions, out=None):
        """K fused steps in ONE kernel launch (per-env state stays in registers).
        actions: [K, N] int32 or [K, N, D] float32, time-major.  Returns (obs, reward, terminated,
        truncated) with a leading K axis; equivalent to K step() calls."""
        K = int(actions.shape[0])
        a = self._as_actions(actions, K)
        if out is None:
            out = self.alloc_rollout(K)
        obs, rew, term, trunc = out
        rc = self._lib.mdpp_step_n(self._h, K, C.c_void_p(a.data_ptr()), C.c_void_p(obs.data_ptr()),
                                   C.c_void_p(rew.data_ptr()), C.c_void_p(term.data_ptr()),
                                   C.c_void_p(trunc.data_ptr()), self._stream())
        capi.check(self._lib, self._h, rc, "mdpp_step_n")
        self._obs_src = obs[K - 1]        # (a view: reset(mask=...) shows it for the envs it leaves alone)
        return obs, rew, term.view(torch.bool), trunc.view(torch.bool)

    def rollout_kernel_name(self, K):
        """Name (with template arguments) of the kernel mdpp_step_n(K) launches for this handle, as
        decided by the library's own dispatch code (mdpp_kernel_name; nothing is launched)."""
        name = self._lib.mdpp_kernel_name(self._h, int(K))
        return name.decode() if name else ""

    def set_kernel_options(self, *disable):
        """Take specialised kernels out of the dispatch (names of capi.OPTIONS, e.g. "NO_PIPE"): the
        handle then runs on the more general kernel of the same arithmetic.  Tests / profiling only;
        results do not depend on it.  No arguments = default dispatch."""
        mask = 0
        for d in disable:
            mask |= capi.OPTIONS[d]
        capi.check(self._lib, self._h, self._lib.mdpp_set_options(self._h, mask), "mdpp_set_options")

    def alloc_rollout(self, K):
        N, dev = self.num_envs, self.device
        shape = self._obs_shape(K, N)
        return (torch.empty(shape, dtype=self._obs_torch_dtype, device=dev),
                torch.empty((K, N), dtype=torch.float32, device=dev),
                torch.empty((K, N), dtype=torch.uint8, device=dev),
                torch.empty((K, N), dtype=torch.uint8, device=dev))

    def _as_actions(self, actions, K):
        want = torch.float32 if self.kind == "continuous" else torch.int32
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        if self.kind == "continuous" and actions.dtype != torch.float32:
            # the reference rejects non-float32 actions (Box.contains -> "stay", :1640,:1671);
            # a batched API cannot flag dtype per env, so raise instead of silently staying
            raise TypeError(f"continuous actions must be float32, got {actions.dtype}")
        a = actions.to(device=self.device, dtype=want).contiguous()
        lead = (self.num_envs,) if K is None else (K, self.num_envs)
        if self.kind == "discrete":
            shape = lead + (2,) if self._irr else lead
        elif self.kind == "grid":
            # the reference treats a non-integer action as outside the action space (noop, :1731);
            # a batched API cannot flag dtype per env, so raise instead of silently staying
            if actions.dtype.is_floating_point:
                raise TypeError(f"grid actions must be integers, got {actions.dtype}")
            shape = lead + (len(self.mdps[0].grid_shape),)
        else:
            shape = lead + (self.mdps[0].D,)
        if tuple(a.shape) != shape:
            raise ValueError(f"actions must have shape {shape}, got {tuple(a.shape)}")
        return a

    # ------------------------------------------------------------------ state access
    def get_augmented_state(self):
        """Batched get_augmented_state() (rl_toy_env.py:2127) plus what the reference leaves out:
        reward ring, step counters, reached flags, and with autoreset="next_step" the per-env flag
        "the next call is the reset" (`reset_pending`).  Host numpy arrays (synchronises)."""
        st = self._get_augmented_state()
        if self.autoreset == "next_step":
            pend = np.zeros(self.num_envs, np.uint8)
            rc = self._lib.mdpp_get_reset_pending(self._h, capi.nptr(pend))
            capi.check(self._lib, self._h, rc, "mdpp_get_reset_pending")
            st["reset_pending"] = pend.astype(bool)
        return st

    def _get_augmented_state(self):
        N = self.num_envs
        if self.kind == "discrete":
            L, d = self._cfg.L, self._cfg.delay
            hist = np.zeros((N, L + 1), np.int32)
            steps = np.zeros(N, np.int32)
            ring = np.zeros((N, d), np.float64)
            rc = self._lib.mdpp_get_state_discrete(self._h, capi.nptr(hist), capi.nptr(steps), capi.nptr(ring))
            capi.check(self._lib, self._h, rc, "mdpp_get_state_discrete")
            cur = hist[:, -1].astype(np.int64)
            if self._irr:                              # curr_state = (relevant, irrelevant), :2089
                irr = np.zeros(N, np.int32)
                rc = self._lib.mdpp_get_state_irrelevant(self._h, capi.nptr(irr))
                capi.check(self._lib, self._h, rc, "mdpp_get_state_irrelevant")
                cur = np.stack([cur, irr.astype(np.int64)], axis=1)
            return {"curr_state": cur, "curr_obs": cur,
                    "augmented_state": hist, "total_transitions_episode": steps, "reward_buffer": ring}
        if self.kind == "grid":
            G = self._cfg.grid_dims
            cells = np.zeros((N, G), np.int32)
            steps = np.zeros(N, np.int32)
            reached = np.zeros(N, np.uint8)
            rc = self._lib.mdpp_get_state_grid(self._h, capi.nptr(cells), capi.nptr(steps), capi.nptr(reached))
            capi.check(self._lib, self._h, rc, "mdpp_get_state_grid")
            cur = cells.astype(np.int64)
            return {"curr_state": cur, "curr_obs": cur, "augmented_state": cur[:, :2],
                    "total_transitions_episode": steps, "reached_terminal": reached.astype(bool)}
        D, n, d = self._cfg.D, self._cfg.order, self._cfg.delay
        sd = np.zeros((N, n + 1, D), np.float32)
        cur = np.zeros((N, D), np.float32)
        steps = np.zeros(N, np.int32)
        ring = np.zeros((N, d), np.float64)
        is32 = np.zeros((N, d), np.uint8)
        reached = np.zeros(N, np.uint8)
        rc = self._lib.mdpp_get_state_continuous(self._h, capi.nptr(sd), capi.nptr(cur), capi.nptr(steps),
                                                 capi.nptr(ring), capi.nptr(is32), capi.nptr(reached))
        capi.check(self._lib, self._h, rc, "mdpp_get_state_continuous")
        aug = cur
        if self._line_L:
            # move_along_a_line: what the reference's augmented_state is read for (rl_toy_env.py:1865-1872, :2147-2156) --
            # the last sequence_length states' relevant coordinates, oldest first, NaN before the episode's reset
            aug = np.zeros((N, self._line_L, self._cfg.n_rel), np.float32)
            rc = self._lib.mdpp_get_line_history(self._h, capi.nptr(aug))
            capi.check(self._lib, self._h, rc, "mdpp_get_line_history")
        return {"curr_state": cur, "curr_obs": cur, "augmented_state": aug, "state_derivatives": sd,
                "total_transitions_episode": steps, "reward_buffer": ring, "reward_buffer_is32": is32,
                "reached_terminal": reached.astype(bool)}

    def set_augmented_state(self, state):
        """Inverse of get_augmented_state() (rl_toy_env.py:2168).  `reset_pending` (next-step autoreset) is
        restored when present and cleared when absent."""
        self._set_augmented_state(state)
        pend = state.get("reset_pending") if isinstance(state, dict) else None
        pend = np.zeros(self.num_envs, np.uint8) if pend is None else np.ascontiguousarray(pend, dtype=np.uint8)
        if pend.shape != (self.num_envs,):
            raise ValueError("reset_pending must have shape (num_envs,)")
        if self.autoreset == "next_step" or pend.any():
            rc = self._lib.mdpp_set_reset_pending(self._h, capi.nptr(pend))
            capi.check(self._lib, self._h, rc, "mdpp_set_reset_pending")

    def _set_augmented_state(self, state):
        if self.kind == "discrete":
            hist = np.ascontiguousarray(state["augmented_state"], dtype=np.int32)
            steps = np.ascontiguousarray(state["total_transitions_episode"], dtype=np.int32)
            ring = state.get("reward_buffer")       # (absent: the delay line stays as it is, like the reference)
            ring = None if ring is None else np.ascontiguousarray(ring, np.float64)
            rc = self._lib.mdpp_set_state_discrete(self._h, capi.nptr(hist), capi.nptr(steps), capi.nptr(ring))
            capi.check(self._lib, self._h, rc, "mdpp_set_state_discrete")
            if self._irr:
                irr = np.ascontiguousarray(np.asarray(state["curr_state"])[:, 1], dtype=np.int32)
                rc = self._lib.mdpp_set_state_irrelevant(self._h, capi.nptr(irr))
                capi.check(self._lib, self._h, rc, "mdpp_set_state_irrelevant")
            return
        if self.kind == "grid":
            cells = np.ascontiguousarray(state["curr_state"], dtype=np.int32)
            steps = np.ascontiguousarray(state["total_transitions_episode"], dtype=np.int32)
            reached = state.get("reached_terminal")
            reached = None if reached is None else np.ascontiguousarray(reached, np.uint8)
            rc = self._lib.mdpp_set_state_grid(self._h, capi.nptr(cells), capi.nptr(steps), capi.nptr(reached))
            capi.check(self._lib, self._h, rc, "mdpp_set_state_grid")
            return
        sd = np.ascontiguousarray(state["state_derivatives"], dtype=np.float32)
        cur = np.ascontiguousarray(state["curr_state"], dtype=np.float32)
        steps = np.ascontiguousarray(state["total_transitions_episode"], dtype=np.int32)
        ring = state.get("reward_buffer")
        ring = None if ring is None else np.ascontiguousarray(ring, np.float64)
        is32 = state.get("reward_buffer_is32")
        is32 = None if is32 is None else np.ascontiguousarray(is32, np.uint8)
        reached = state.get("reached_terminal")
        reached = None if reached is None else np.ascontiguousarray(reached, np.uint8)
        rc = self._lib.mdpp_set_state_continuous(self._h, capi.nptr(sd), capi.nptr(cur), capi.nptr(steps),
                                                 capi.nptr(ring), capi.nptr(is32), capi.nptr(reached))
        capi.check(self._lib, self._h, rc, "mdpp_set_state_continuous")
        if self._line_L:
            # two layouts: the [N, sequence_length, n_relevant] window get_augmented_state() returns here, or the REFERENCE's
            # (rl_toy_env.py:660, :2147-2156): per env a list of sequence_length + delay + 1 full state vectors, oldest first
            # -- [N, L + delay + 1, D]; the reward reads its last L rows and the relevant columns (:1865-1872), which is what
            # is kept (the rows before them only feed the delay line, restored through `reward_buffer`)
            aug = np.asarray(state["augmented_state"], dtype=np.float32)
            D, L, d = self._cfg.D, self._line_L, self._cfg.delay
            if aug.shape == (self.num_envs, L + d + 1, D):
                rel = list(self.mdps[0].relevant_indices)
                aug = aug[:, -L:, :][:, :, rel]
            aug = np.ascontiguousarray(aug, dtype=np.float32)
            if aug.shape != (self.num_envs, L, self._cfg.n_rel):
                raise ValueError("move_along_a_line: augmented_state must be the [num_envs, sequence_length, n_relevant] history "
                                 "get_augmented_state() returns, or the reference's [num_envs, sequence_length + delay + 1, "
                                 "state_space_dim] list of states")
            rc = self._lib.mdpp_set_line_history(self._h, capi.nptr(aug))
            capi.check(self._lib, self._h, rc, "mdpp_set_line_history")

    def get_episode_stats(self):
        """The reference's per-episode statistics (attributes of every env object, logged at each reset() and cleared,
        rl_toy_env.py:2231-2247, :2360-2369), per env instance, for handles made with ``episode_stats=True``: a dict
        with the running episode's ``total_abs_noise_in_reward_episode``, ``total_reward_episode``,
        ``total_noisy_transitions_episode`` (discrete, grid), ``total_abs_noise_in_transition_episode`` ([N, D],
        continuous), ``total_transitions_episode``, and under ``"last_episode"`` the same for the episode each env's
        latest reset() ended (what the reference logs).  Host numpy arrays (synchronises)."""
        if not self.episode_stats:
            raise capi.MdppError("get_episode_stats: construct the env with episode_stats=True")
        N = self.num_envs
        D = self.mdps[0].D if self.kind == "continuous" else 0
        nk = 3 + D
        cur = np.zeros((nk, N), np.float64)
        last = np.zeros((nk + 1, N), np.float64)
        rc = self._lib.mdpp_get_episode_stats(self._h, capi.nptr(cur), capi.nptr(last))
        capi.check(self._lib, self._h, rc, "mdpp_get_episode_stats")

        def rows(a, transitions):
            d = {"total_abs_noise_in_reward_episode": a[0].copy(), "total_reward_episode": a[1].copy(),
                 "total_transitions_episode": transitions}
            if self.kind == "continuous":
                # (total_reward_episode of a continuous env is the reference's np.float32 running sum wherever its reward
                #  is np.float32 -- held here as the float64 of exactly that value)
                d["total_abs_noise_in_transition_episode"] = np.ascontiguousarray(a[3:3 + D].T)
            else:
                d["total_noisy_transitions_episode"] = a[2].astype(np.int64)
            return d
        out = rows(cur, np.asarray(self._get_augmented_state()["total_transitions_episode"], dtype=np.int64))
        out["last_episode"] = rows(last, last[nk].astype(np.int64))
        return out

    def get_rng_streams(self, stream=capi.STREAM_ENV):
        words = np.zeros((self.num_envs, 6), np.uint64)
        rc = self._lib.mdpp_get_streams(self._h, stream, capi.nptr(words))
        capi.check(self._lib, self._h, rc, "mdpp_get_streams")
        return words

    def status(self):
        """Per-env sticky fault bits (bad action, ...), cleared by the call."""
        flags = np.zeros(self.num_envs, np.uint32)
        rc = self._lib.mdpp_status(self._h, capi.nptr(flags))
        capi.check(self._lib, self._h, rc, "mdpp_status")
        return flags

    # kernel timing on the caller's stream (HIP events inside the library)
    def timer_begin(self):
        capi.check(self._lib, self._h, self._lib.mdpp_timer_begin(self._h, self._stream()), "mdpp_timer_begin")

    def timer_end(self):
        ms = C.c_float()
        capi.check(self._lib, self._h, self._lib.mdpp_timer_end(self._h, self._stream(), C.byref(ms)), "mdpp_timer_end")
        return ms.value

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize(self.device)
            self._lib.mdpp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StepGraph:
    """What RLToyVectorEnv.step_graph() returns: a captured HIP graph of K mdpp_step launches.  replay() runs
    the K steps on the current stream and advances the handle's step counter by K, as K step() calls would."""

    def __init__(self, env, graph, K, tick0, actions, obs, reward, term, trunc, by_offset=False):
        self._env, self._g, self.K, self._tick0, self._by_offset = env, graph, K, tick0, by_offset
        self.actions, self.obs, self.reward = actions, obs, reward
        self.terminated, self.truncated = term.view(torch.bool), trunc.view(torch.bool)

    def replay(self):
        env = self._env
        d = int(env._cfg.delay)
        now = C.c_uint64()
        capi.check(env._lib, env._h, env._lib.mdpp_tick(env._h, 0, C.byref(now)), "mdpp_tick")
        if self._by_offset:
            # the captured launches add this to the counter they were captured with (in stream order before the graph)
            stream = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
            capi.check(env._lib, env._h, env._lib.mdpp_graph_set_tick_offset(env._h, int(now.value) - self._tick0, stream),
                       "mdpp_graph_set_tick_offset")
        else:
            ring_in_memory = d > 0 and (env.kind == "continuous" or (env.kind == "discrete" and not env._cfg.unit_rewards))
            if ring_in_memory and (int(now.value) - self._tick0) % d != 0:
                raise capi.MdppError("StepGraph.replay: %d steps were taken outside the graph since its capture; the delay line "
                                     "(length %d) would be read at the captured ring head" % (int(now.value) - self._tick0, d))
        self._g.replay()
        capi.check(env._lib, env._h, env._lib.mdpp_tick(env._h, self.K, None), "mdpp_tick")
        env._obs_src = self.obs[self.K - 1]


def make_vec(env_id="RLToyVec-v0", num_envs=1, **kwargs):
    """gym.make-style entry: 'RLToyVec-v0' and 'RLToyVecFiniteHorizon-v0' (max_episode_steps=100),
    the batched counterparts of RLToy-v0 / RLToyFiniteHorizon-v0 (mdp_playground/__init__.py:3-12)."""
    if env_id == "RLToyVec-v0":
        return RLToyVectorEnv(num_envs=num_envs, **kwargs)
    if env_id == "RLToyVecFiniteHorizon-v0":
        kwargs.setdefault("max_episode_steps", 100)
        return RLToyVectorEnv(num_envs=num_envs, **kwargs)
    raise ValueError(f"unknown env id {env_id}")
