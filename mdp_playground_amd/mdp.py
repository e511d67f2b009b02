"""Host-side MDP generation for the batched RLToyEnv.

Restates what ``RLToyEnv.__init__`` does before the first ``step()`` —
/root/reference/mdp_playground/envs/rl_toy_env.py:216-853 (defaults :342-566,
``init_terminal_states`` :855, ``init_init_state_dist`` :992,
``init_transition_function`` :1042, ``init_reward_function`` :1253) — and turns the
result into flat tables the HIP kernels consume.  It draws from numpy
``Generator(PCG64)`` objects in exactly the order the reference does, so for the same
config/seed the tables and the post-construction RNG states are identical to the
reference's (pinned by tests/test_mdp_builder.py against tests/golden/*.npz).

None of this is on the per-step hot path; it runs once per MDP on the host.
"""
from __future__ import annotations

import copy
import sys
import warnings
from dataclasses import dataclass, field

import numpy as np

_SEED_KEYS = ("relevant_state_space", "relevant_action_space", "irrelevant_state_space",
              "irrelevant_action_space", "state_space", "action_space",
              "image_representations")


def new_generator(seed):
    """gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed)))."""
    if seed is not None and not (isinstance(seed, int) and seed >= 0):
        raise TypeError(f"Seed must be a non-negative python int or None, got {seed!r}")
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


def pcg64_words(gen) -> np.ndarray:
    """uint64[6] = {state_lo, state_hi, inc_lo, inc_hi, has_uint32, uinteger} of a PCG64 Generator."""
    st = gen.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s & m, s >> 64, inc & m, inc >> 64, st["has_uint32"], st["uinteger"]],
                    dtype=np.uint64)


_FRESH_CACHE = {}          # (seed, count, stride) -> words: seeding 65 536 generators takes 0.25 s, and batches of one seed recur


def fresh_stream_words(seed, count, stride=1) -> np.ndarray:
    """PCG64 words of ``count`` freshly seeded generators with seeds seed, seed+stride, ...  (a pure function of its
    arguments for a seed that is not None: the last few results are kept and handed out as copies)"""
    key = (int(seed), int(count), int(stride)) if seed is not None else None
    if key is not None and key in _FRESH_CACHE:
        return _FRESH_CACHE[key].copy()
    out = np.empty((count, 6), dtype=np.uint64)
    for i in range(count):
        out[i] = pcg64_words(new_generator(None if seed is None else seed + i * stride))
    if key is not None and count >= 1024:
        if len(_FRESH_CACHE) >= 8:
            _FRESH_CACHE.pop(next(iter(_FRESH_CACHE)))
        _FRESH_CACHE[key] = out.copy()
    return out


@dataclass
class CommonParams:
    delay: int = 0
    sequence_length: int = 1
    reward_every_n_steps: int = 1
    reward_noise: float | None = None       # std; None = no draw at all
    reward_scale: float = 1.0
    reward_shift: float = 0.0
    term_state_reward: float = 0.0
    seed_dict: dict = field(default_factory=dict)


@dataclass
class DiscreteMDP(CommonParams):
    kind: str = "discrete"
    S: int = 0
    A: int = 0
    diameter: int = 1
    transition_noise: float | None = None
    P: np.ndarray = None                     # int64 [S, A]
    terminal_states: np.ndarray = None       # int64 [n_term]
    init_dist: np.ndarray = None             # float64 [S]
    rewardable_sequences: dict = None        # tuple(states) -> float (incl. make_denser sub-sequences)
    space_rng_words: np.ndarray = None       # observation_spaces[0] RNG after P generation
    image: dict | None = None                # ImageMultiDiscrete parameters, or None
    # irrelevant sub-space (irrelevant_features=True: Tuple spaces, rl_toy_env.py:2028-2092)
    irrelevant: bool = False
    S_irr: int = 0
    A_irr: int = 0
    P_irr: np.ndarray = None                 # int64 [S_irr, A_irr]
    init_dist_irr: np.ndarray = None         # float64 [S_irr], uniform (:1025-1037)
    space_irr_rng_words: np.ndarray = None   # observation_spaces[1] RNG after P generation
    space_seeds: tuple = None                # seeds the two sub-space generators were made from
    # use_custom_mdp with matrices (rl_toy_env.py:1232-1236, :1259-1267): R(s, a), float64 [S, A]
    reward_matrix: np.ndarray = None

    def reward_table(self) -> np.ndarray:
        """Dense float64[S**L]; key(seq) = sum seq[i] * S**(L-1-i).  Only full-length keys can
        match in step() (rl_toy_env.py:1837-1841), shorter make_denser keys never do.
        With a custom reward matrix: float64[S*A], key = s * A + a."""
        if self.reward_matrix is not None:
            return np.ascontiguousarray(self.reward_matrix, dtype=np.float64).ravel()
        L, S = self.sequence_length, self.S
        t = np.zeros(S ** L, dtype=np.float64)
        for seq, val in self.rewardable_sequences.items():
            if len(seq) != L:
                continue
            k = 0
            for s in seq:
                k = k * S + int(s)
            t[k] = val
        return t

    def is_terminal_table(self) -> np.ndarray:
        t = np.zeros(self.S, dtype=np.uint8)
        t[np.asarray(self.terminal_states, dtype=np.int64)] = 1
        return t

    def init_cdf(self) -> np.ndarray:
        cdf = np.cumsum(np.asarray(self.init_dist, dtype=np.float64))
        return cdf / cdf[-1]

    def init_cdf_irr(self) -> np.ndarray:
        cdf = np.cumsum(np.asarray(self.init_dist_irr, dtype=np.float64))
        return cdf / cdf[-1]

    def noise_cdf(self, irrelevant=False) -> np.ndarray | None:
        """Row n = normalised cdf Generator.choice builds for the P-noise categorical with mode n
        (rl_toy_env.py:1605-1612; irrelevant sub-space :2068-2076)."""
        if not self.transition_noise:
            return None
        S = self.S_irr if irrelevant else self.S
        out = np.empty((S, S), dtype=np.float64)
        for n in range(S):
            probs = np.ones(shape=(S,)) * self.transition_noise / (S - 1)
            probs[n] = 1 - self.transition_noise
            cdf = probs.cumsum()
            cdf /= cdf[-1]
            out[n] = cdf
        return out


@dataclass
class ContinuousMDP(CommonParams):
    kind: str = "continuous"
    D: int = 0
    relevant_indices: list = None
    order: int = 1
    inertia: float = 1.0
    time_unit: float = 1.0
    state_space_max: float = np.inf
    action_space_max: float = np.inf
    target_point: np.ndarray = None          # float32 [n_rel]
    target_default: bool = False             # no target_point in the config: float64 zeros (:652-654), float64 reward path
    target_radius: float = 0.05
    make_denser: bool = True
    action_loss_weight: float = 0.0
    transition_noise: float | None = None    # std; None = no draw
    box_lo: np.ndarray = None                # float32 [K, n_rel] terminal hypercubes
    box_hi: np.ndarray = None
    image: dict | None = None                # ImageContinuous parameters (width, height, circle_radius), or None
    reward_function: str = "move_to_a_point"  # or "move_along_a_line" (:1864-1910)


@dataclass
class GridMDP(CommonParams):
    kind: str = "grid"
    grid_shape: tuple = ()                   # doubled with irrelevant_features (:604-608)
    n_rel: int = 2                           # leading dimensions the reward looks at
    target_point: list = None
    make_denser: bool = False
    transition_noise: float | None = None    # probability of replacing the action (:1736)
    terminal_states: list = None             # kept for the record: they never terminate, see build_grid
    image: dict | None = None                # ImageContinuous parameters (width, height, circle_radius), or None


def _require(cond, msg):
    if not cond:
        raise AssertionError(msg)


def _seed_dict(config):
    """rl_toy_env.py:285-333: an int seed spawns 7 sub-seeds from the env generator."""
    if "seed" not in config or config["seed"] is None:
        seed_int = None
    elif isinstance(config["seed"], dict):
        sd = dict(config["seed"])
        return sd, new_generator(sd["env"])
    elif isinstance(config["seed"], int):
        seed_int = config["seed"]
    else:
        raise TypeError("Unsupported data type for seed, actual config: ",
                        type(config["seed"]), config)
    env_rng = new_generator(seed_int)
    sd = {"env": seed_int}
    for k in _SEED_KEYS:
        sd[k] = env_rng.integers(sys.maxsize).item()
    return sd, env_rng


def _common(config, kind, sd):
    L = config.get("sequence_length", 1)
    _require(L > 0, 'config["sequence_length"] <= 0. Set to: ' + str(L))
    every_n = config.get("reward_every_n_steps", L if kind == "discrete" else 1)
    rn = config.get("reward_noise", None)
    if callable(rn):
        raise NotImplementedError("callable reward_noise is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a float std (reference rl_toy_env.py:398-403)")
    return dict(delay=config.get("delay", 0), sequence_length=L, reward_every_n_steps=every_n,
                reward_noise=None if rn is None else float(rn),
                reward_scale=config.get("reward_scale", 1.0),
                reward_shift=config.get("reward_shift", 0.0),
                term_state_reward=config.get("term_state_reward", 0.0), seed_dict=sd)


def _sample_space(rng, n, prob=None, size=1, replace=True):
    """DiscreteExtended.sample (spaces/discrete_extended.py:11-23)."""
    sampled = np.squeeze(rng.choice(n, size=size, p=prob, replace=replace))
    return int(sampled) if sampled.shape == () else sampled


def _next_set_prob(S, A, s):
    """Uniform mass on the independent set following the one s lives in (:1074-1092)."""
    i_s = s // A
    prob = np.zeros(shape=(S,))
    ind_1 = ((i_s + 1) * A) % S
    ind_2 = ((i_s + 2) * A) % S
    if ind_2 <= ind_1:
        ind_2 += S
    prob[ind_1:ind_2] = np.ones(shape=(A,)) / A
    return prob


def _rewardable_sequences(env_rng, n_nonterm, A, L, fraction, repeats, diameter):
    """get_sequences, rl_toy_env.py:1273-1473: pick sequence numbers without replacement from
    the env generator and decode them (base-n digits with repeats, a Lehmer-style mixed-radix
    code without)."""
    seqs = []
    if repeats:
        total = n_nonterm ** L
        n_sel = int(fraction * total)
        if n_sel == 0:
            n_sel = 1
            warnings.warn("0 rewardable sequences per independent set for given reward_density, "
                          "sequence_length, diameter and terminal_state_density. Setting it to 1.")
        picks = env_rng.choice(total, size=n_sel, replace=False)
        for i_s in range(diameter):
            for num in picks:
                seq = []
                while len(seq) != L:
                    seq.append(num % n_nonterm + ((len(seq) + i_s) % diameter) * A)
                    num = num // n_nonterm
                seqs.append(seq)
        return seqs
    _require(L <= diameter * n_nonterm, "When there are no repeats in sequences, the sequence "
             "length should be <= diameter * maximum.")
    radices = [n_nonterm - (i // diameter) for i in range(L)]
    for i_s in range(diameter):
        total = np.prod(radices)
        n_sel = int(fraction * total)
        if n_sel == 0:
            n_sel = 1
            warnings.warn("0 rewardable sequences per independent set for given reward_density, "
                          "sequence_length, diameter and terminal_state_density. Setting it to 1.")
        picks = env_rng.choice(total, size=n_sel, replace=False)
        for num in picks:
            pools = [list(range(n_nonterm)) for _ in range(diameter)]
            seq = []
            for pos, radix in enumerate(radices):
                which = (pos + i_s) % diameter
                digit = num % radix
                seq.append(pools[which].pop(digit) + which * A)
                num = num // radix
            _require(seq not in seqs, "None of the generated sequences should have clashed with "
                     "an existing rewardable sequence when it was generated.")
            seqs.append(seq)
    return seqs


def build_discrete(config) -> DiscreteMDP:
    config = copy.deepcopy(config)
    if config.get("use_custom_mdp", False):
        return _build_discrete_custom(config)
    sd, env_rng = _seed_dict(config)
    common = _common(config, "discrete", sd)
    L = common["sequence_length"]
    diameter = config.get("diameter", 1)
    irrelevant = bool(config.get("irrelevant_features", False))
    A_irr = 0
    if irrelevant:
        _require(len(config["action_space_size"]) == 2,
                 "Currently, 1st sub-state (and action) space is assumed to be relevant to rewards "
                 "and 2nd one is irrelevant. Please provide a list with sizes for the 2.")
        A, A_irr = (int(x) for x in config["action_space_size"])
    else:
        _require(isinstance(config["action_space_size"], int),
                 "Did you mean to turn irrelevant_features? If so, please set irrelevant_features = "
                 "True in config. If not, please provide an int for action_space_size.")
        A = config["action_space_size"]
    S = A * diameter                                      # :589-591
    S_irr = A_irr * diameter
    tn = config.get("transition_noise", None)
    # :868-881 terminal states = the last int(density * A) states of every independent set
    n_term = int(config.get("terminal_state_density", 0.25) * A)
    terminal = np.array([j * A - 1 - i for j in range(1, diameter + 1) for i in range(n_term)],
                        dtype=np.int64)
    n_nonterm = A - n_term
    # :1003-1018 rho_0 uniform over non-terminal states
    init_dist = np.array(([1 / (n_nonterm * diameter) for _ in range(n_nonterm)]
                          + [0 for _ in range(n_term)]) * diameter)
    # :1050-1151 P, drawn from the relevant state space's own generator.  With an irrelevant
    # sub-space the two state spaces are wrapped in a TupleExtended seeded with
    # seed_dict["state_space"] (:725-728), and gymnasium's Tuple.seed(int) (third-party, 0.29 / 1.x)
    # re-seeds every sub-space: subseeds = Generator(seed).integers(int32 max, size=len(spaces)).
    # With image observations the observation space is the ImageMultiDiscrete instead (:707-717): no
    # Tuple is built, and the two state spaces keep the seeds they were constructed with.
    space_seeds = (sd["relevant_state_space"], sd.get("irrelevant_state_space"))
    if irrelevant and not config.get("image_representations", False):
        subseeds = new_generator(sd["state_space"]).integers(np.iinfo(np.int32).max, size=2)
        space_seeds = (int(subseeds[0]), int(subseeds[1]))
    space_rng = new_generator(space_seeds[0])
    P = np.full((S, A), -1, dtype=np.int64)
    if config.get("maximally_connected", True):
        for s in range(S):
            if diameter == 1:
                P[s] = _sample_space(space_rng, S, size=A, replace=False)
            else:
                P[s] = _sample_space(space_rng, S, prob=_next_set_prob(S, A, s), size=A,
                                     replace=False)
    else:
        for s in range(S):
            prob = _next_set_prob(S, A, s)
            for a in range(A):
                P[s, a] = _sample_space(space_rng, S, prob=prob)
    for i_s in range(diameter):
        for s in range(A - n_term, A):
            P[i_s * A + s, :] = i_s * A + s               # terminal self-loops, :1135-1151
    P_irr = init_dist_irr = irr_words = None
    if irrelevant:                                        # :1153-1228, always the prob= form
        irr_rng = new_generator(space_seeds[1])
        P_irr = np.full((S_irr, A_irr), -1, dtype=np.int64)
        for s in range(S_irr):
            prob = _next_set_prob(S_irr, A_irr, s)
            if config.get("maximally_connected", True):
                P_irr[s] = _sample_space(irr_rng, S_irr, prob=prob, size=A_irr, replace=False)
            else:
                for a in range(A_irr):
                    P_irr[s, a] = _sample_space(irr_rng, S_irr, prob=prob)
        init_dist_irr = np.array([1 / S_irr for _ in range(S_irr)])     # :1025-1037
        irr_words = pcg64_words(irr_rng)
    # :1508-1558 rewardable sequences, drawn from the env generator
    seqs = _rewardable_sequences(env_rng, n_nonterm, A, L, config.get("reward_density", 0.25),
                                 config.get("repeats_in_sequences", False), diameter)
    reward_dist = config.get("reward_dist", None)
    if isinstance(reward_dist, list):                     # :1528-1544
        n_rews = diameter * len(seqs)
        rews = [1.0] if n_rews == 1 else np.linspace(reward_dist[0], reward_dist[1], num=n_rews)
        _require(rews[-1] == 1.0, "reward_dist interval must end at 1.0")
        env_rng.shuffle(rews)
        reward_dist = lambda rng, r_dict: rews[len(r_dict)]  # noqa: E731
    make_denser = config.get("make_denser", False)
    table = {}
    for seq in seqs:                                      # insert_sequence, :1475-1504
        seq = tuple(int(x) for x in seq)
        table[seq] = reward_dist(env_rng, table) if callable(reward_dist) else 1.0
        if make_denser:
            for n in range(1, len(seq)):
                sub = seq[:n]
                if sub not in table:
                    table[sub] = 0.0
                table[sub] += table[seq] * n / len(seq)
    image = None
    if config.get("image_representations", False):
        image = _image_params(config, sd)
    return DiscreteMDP(kind="discrete", S=S, A=A, diameter=diameter,
                       transition_noise=None if not tn else float(tn), P=P,
                       terminal_states=terminal, init_dist=init_dist,
                       rewardable_sequences=table, space_rng_words=pcg64_words(space_rng),
                       image=image, irrelevant=irrelevant, S_irr=S_irr, A_irr=A_irr, P_irr=P_irr,
                       init_dist_irr=init_dist_irr, space_irr_rng_words=irr_words,
                       space_seeds=space_seeds, **common)


def _build_discrete_custom(config) -> DiscreteMDP:
    """use_custom_mdp=True with P and R given as MATRICES (rl_toy_env.py:346-348, :586-587,
    :859-866, :997-1000 + :617-618, :1232-1236, :1259-1267).  Callables are refused (NotImplementedError): a Python
    function cannot be restated as a table lookup, and there is no host path."""
    _require("transition_function" in config, "use_custom_mdp needs transition_function")   # :347
    _require("reward_function" in config, "use_custom_mdp needs reward_function")           # :348
    if callable(config["transition_function"]) or callable(config["reward_function"]):
        raise NotImplementedError("use_custom_mdp with callables is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass transition_function / reward_function as S x A arrays")
    if config.get("irrelevant_features", False):
        # the reference wraps the given state_space_size in a one-element list for custom MDPs (:586-587) and then reads
        # state_space_size[1] for the irrelevant sub-space (:685): RLToyEnv.__init__ itself raises IndexError
        raise IndexError("list index out of range (use_custom_mdp with irrelevant_features: the reference's constructor "
                         "fails the same way, rl_toy_env.py:586-587 / :685)")
    sd, _env_rng = _seed_dict(config)
    common = _common(config, "discrete", sd)
    _require(isinstance(config["action_space_size"], int),
             "Did you mean to turn irrelevant_features? If so, please set irrelevant_features = "
             "True in config. If not, please provide an int for action_space_size.")
    A, S = config["action_space_size"], int(config["state_space_size"])    # :586-587: S is taken as given
    diameter = config.get("diameter", 1)
    P = np.asarray(config["transition_function"])
    R = np.asarray(config["reward_function"], dtype=np.float64)
    if P.shape != (S, A) or R.shape != (S, A):
        raise IndexError("transition_function and reward_function must be state_space_size x action_space_size arrays")
    if P.dtype.kind not in "iu" or P.min() < 0 or P.max() >= S:
        raise IndexError("transition_function entries must be state indices in [0, state_space_size)")
    n_term = int(config.get("terminal_state_density", 0.25) * A)           # :868-870 (A, not S)
    if "terminal_states" in config:                                        # :859-866
        if callable(config["terminal_states"]):
            raise NotImplementedError("callable terminal_states is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a list of states / hypercube centres")
        terminal = np.asarray(config["terminal_states"], dtype=np.int64).reshape(-1)
        terminal = terminal[(terminal >= 0) & (terminal < S)]              # `s in list`: others never match
    else:
        terminal = np.array([j * A - 1 - i for j in range(1, diameter + 1) for i in range(n_term)],
                            dtype=np.int64)
    if "relevant_init_state_dist" in config:                               # :617-618
        init_dist = np.asarray(config["relevant_init_state_dist"], dtype=np.float64)
    elif "init_state_dist" in config:
        init_dist = np.asarray(config["init_state_dist"], dtype=np.float64)
    else:                                                                  # :1003-1018 (sized by A)
        n_nonterm = A - n_term
        init_dist = np.array(([1 / (n_nonterm * diameter) for _ in range(n_nonterm)]
                              + [0 for _ in range(n_term)]) * diameter)
    if init_dist.shape != (S,):
        raise ValueError("'a' and 'p' must have same size")                # what Generator.choice raises in reset()
    tn = config.get("transition_noise", None)
    image = _image_params(config, sd) if config.get("image_representations", False) else None
    return DiscreteMDP(kind="discrete", S=S, A=A, diameter=diameter,
                       transition_noise=None if not tn else float(tn), P=P.astype(np.int64),
                       terminal_states=terminal, init_dist=init_dist, rewardable_sequences={},
                       space_rng_words=pcg64_words(new_generator(sd["relevant_state_space"])),
                       image=image, space_seeds=(sd["relevant_state_space"], sd.get("irrelevant_state_space")),
                       reward_matrix=R, **common)


def _image_params(config, sd):
    """Defaults of rl_toy_env.py:442-496 and the ImageMultiDiscrete ctor call at :707-717."""
    transforms = config.get("image_transforms", "none")
    sh_quant = config.get("image_sh_quant", 1 if "shift" in transforms else None)
    ro_quant = config.get("image_ro_quant", 1 if "rotate" in transforms else None)
    scale_range = config.get("image_scale_range", (0.5, 1.5) if "scale" in transforms else None)
    return dict(width=config.get("image_width", 100), height=config.get("image_height", 100),
                transforms=transforms, sh_quant=sh_quant, ro_quant=ro_quant,
                scale_range=None if scale_range is None else tuple(scale_range),
                circle_radius=20, seed=sd["image_representations"])


def build_continuous(config) -> ContinuousMDP:
    config = copy.deepcopy(config)
    if config.get("use_custom_mdp", False):
        raise NotImplementedError("use_custom_mdp is not on the device path")
    sd, _ = _seed_dict(config)
    common = _common(config, "continuous", sd)
    D = config["state_space_dim"]
    rf = config.get("reward_function", "move_to_a_point")
    if rf not in ("move_to_a_point", "move_along_a_line"):
        raise NotImplementedError("reward_function must be 'move_to_a_point' or 'move_along_a_line'")
    line = rf == "move_along_a_line"
    image = None
    if config.get("image_representations", False):
        # ImageContinuous(feature_space, width, height, term_spaces, target_point, circle_radius=5)
        # (:770-778): RGB, relevant_indices left at its default [0, 1], at most 2 + 2 dimensions
        image = dict(width=config.get("image_width", 100), height=config.get("image_height", 100),
                     circle_radius=5)
    if not line:
        _require(common["sequence_length"] == 1, "move_to_a_point needs sequence_length == 1")   # :643
    if config.get("irrelevant_features", False):
        _require("relevant_indices" in config,
                 "Please provide dimensions of state space relevant to rewards.")
    rel = list(config.get("relevant_indices", range(D)))
    if image is not None and (D not in (2, 4) or len(rel) != 2):
        raise NotImplementedError("ImageContinuous observations: 2 relevant dimensions, and 2 or 4 state "
                                  "dimensions in all (the reference's picture is built from dims [0, 1] and [2, 3])")
    if line:
        # :1864-1910: the fit runs on the device -- a scatter matrix kept in registers for up to 8 relevant dimensions (4 x 4, or
        # 8 x 8 in its own kernel instantiation for 5 to 8 of at most 12 state dimensions), in an HBM workspace beyond that
        # (c_line_reward_big) -- over up to 64 states; no target, no target latch (:1719)
        if common["sequence_length"] > 64:
            raise NotImplementedError("move_along_a_line: sequence_length <= 64 (the window of the line fit kept per env)")
        if image is not None:     # :767-775 hands self.target_point to ImageContinuous; only move_to_a_point sets it (:650)
            raise AttributeError("'RLToyEnv' object has no attribute 'target_point' (the reference's constructor fails for "
                                 "move_along_a_line with image_representations)")
        target = np.zeros(len(rel), dtype=np.float32)
    elif "target_point" in config:
        target = np.array(config["target_point"], dtype=np.float32)
        _require(target.shape == (len(rel),),
                 "target_point should have dimensionality = relevant_state_space dimensionality")
    else:
        # The reference's default (:652-654): np.zeros(shape=(state_space_dim,)) -- FLOAT64 and of the full state
        # dimension.  `state[relevant_indices] - target_point` then only broadcasts when every dimension is relevant
        # (the reference raises ValueError at its first step otherwise), and it is a float64 vector: distances, the
        # target latch and a dense reward are float64 (ContinuousMDP.target_default; kernels: target64).
        if len(rel) != D and len(rel) != 1:
            raise ValueError("operands could not be broadcast together: the default target_point has state_space_dim = "
                             f"{D} entries, the relevant state {len(rel)} (pass target_point)")
        if len(rel) == 1 and D != 1:
            raise NotImplementedError("default target_point with one relevant dimension of several (the reference "
                                      "broadcasts it against all state_space_dim zeros): pass target_point")
        if D >= 16:
            raise NotImplementedError("default (float64) target_point: state_space_dim < 16 on the device "
                                      "(numpy's float64 dot changes its summation order from 16 elements on)")
        if image is not None:
            raise NotImplementedError("default target_point with image observations: pass target_point")
        target = np.zeros(len(rel), dtype=np.float32)
        target_default = True
    if not (not line and "target_point" not in config):
        target_default = False
    tn = config.get("transition_noise", None)
    if callable(tn):
        raise NotImplementedError("callable transition_noise is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a float")
    box_lo = box_hi = None
    if "terminal_states" in config:
        if callable(config["terminal_states"]):
            raise NotImplementedError("callable terminal_states is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a list of states / hypercube centres")
        ts = config["terminal_states"]
        for i, c in enumerate(ts):
            _require(len(c) == len(rel), "Specified terminal state centres should have "
                     "dimensionality = number of relevant_indices. That was not the case for "
                     "centre no.: " + str(i))
        edge = config["term_state_edge"]
        box_lo = np.array([[c[j] - edge / 2 for j in range(len(rel))] for c in ts]).astype(np.float32)
        box_hi = np.array([[c[j] + edge / 2 for j in range(len(rel))] for c in ts]).astype(np.float32)
    return ContinuousMDP(
        kind="continuous", D=D, relevant_indices=rel,
        order=config.get("transition_dynamics_order", 1), inertia=config.get("inertia", 1.0),
        time_unit=config.get("time_unit", 1.0),
        state_space_max=config.get("state_space_max", np.inf),
        action_space_max=config.get("action_space_max", np.inf),
        target_point=target, target_default=target_default, target_radius=config.get("target_radius", 0.05),
        make_denser=config.get("make_denser", True),
        action_loss_weight=config.get("action_loss_weight", 0.0),
        transition_noise=None if tn is None else float(tn), box_lo=box_lo, box_hi=box_hi, image=image,
        reward_function=rf, **common)


def build_grid(config) -> GridMDP:
    """Grid envs, rl_toy_env.py:539-541, :604-608, :655-657, :780-811, :958-987.  What the reference
    actually supports is narrower than its docstring, and the limits are kept:
      * exactly 2 grid dimensions (the augmented state keeps 2 coordinates, :2055-2056, and the
        reward subtracts the target from them, :1949-1958);
      * delay 0 and sequence_length 1 (otherwise reward_function builds a ragged np.array out of
        NaN placeholders and coordinate lists and raises, :1949);
      * list-form terminal_states never terminate an episode: they become Box(dtype=int64) spaces
        and is_terminal_state() asks them whether a float64 array is contained (:973-982), which
        gymnasium's Box.contains answers with False for any non-castable dtype.  Only reaching
        the target ends an episode (:1770-1776)."""
    config = copy.deepcopy(config)
    if config.get("use_custom_mdp", False):
        raise NotImplementedError("use_custom_mdp is not on the device path")
    sd, _env_rng = _seed_dict(config)
    common = _common(config, "grid", sd)
    _require("grid_shape" in config, "grid envs need grid_shape")
    shape = tuple(int(g) for g in config["grid_shape"])
    if config.get("reward_function") != "move_to_a_point":
        raise NotImplementedError("grid envs: only reward_function='move_to_a_point' exists (:79)")
    if "make_denser" not in config:
        raise AttributeError("'RLToyEnv' object has no attribute 'make_denser'")   # what the reference raises (:1949)
    if len(shape) != 2:
        raise NotImplementedError("the reference's grid reward only works for 2-D grids (:1949-1958 broadcast error)")
    if common["delay"] != 0 or common["sequence_length"] != 1:
        raise NotImplementedError("the reference's grid reward raises for delay > 0 or sequence_length > 1 (:1949)")
    if callable(config.get("terminal_states")):
        raise NotImplementedError("callable terminal_states is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a list of states / hypercube centres")
    image = None
    if config.get("image_representations", False):
        # ImageContinuous(feature_space, width, height, term_spaces, target_point, circle_radius=5,
        # grid_shape=...) (:800-811): RGB, grid lines, terminal cells drawn (though they never terminate)
        image = dict(width=config.get("image_width", 100), height=config.get("image_height", 100),
                     circle_radius=5)
        if len(config.get("terminal_states") or []) > 128:
            raise NotImplementedError("at most 128 terminal cells are drawn on the device")
    tn = config.get("transition_noise", None)
    if callable(tn):
        raise NotImplementedError("callable transition_noise is not supported: a Python function cannot be evaluated on the device, and no host path is shipped; pass a float")
    target = [int(x) for x in config["target_point"]]
    _require(len(target) == 2, "target_point must have one coordinate per grid dimension")
    if max(shape) > 254:
        raise NotImplementedError("grid sides up to 254 cells")
    irr = bool(config.get("irrelevant_features", False))
    return GridMDP(kind="grid", grid_shape=shape * 2 if irr else shape, n_rel=2, target_point=target,
                   make_denser=bool(config["make_denser"]),
                   transition_noise=None if not tn else float(tn),
                   terminal_states=config.get("terminal_states"), image=image, **common)


def build_mdp(config):
    """Dispatch on state_space_type like rl_toy_env.py:339,499-543."""
    if config == {}:
        config = {"state_space_size": 8, "action_space_size": 8, "state_space_type": "discrete",
                  "action_space_type": "discrete", "terminal_state_density": 0.25,
                  "maximally_connected": True}
    kind = config["state_space_type"].lower()
    if kind == "discrete":
        return build_discrete(config)
    if kind == "continuous":
        return build_continuous(config)
    if kind == "grid":
        return build_grid(config)
    raise ValueError("Unknown state_space_type")


def _build_chunk(args):
    config, seeds = args
    return [build_mdp({**config, "seed": s}) for s in seeds]


def build_many(config, seeds, workers=None):
    """[build_mdp({**config, "seed": s}) for s in seeds] -- one MDP per env instance (RLToyVectorEnv(seeds=...)) -- built by a
    pool of forked worker processes: 65 536 discrete 8 x 8 MDPs take 0.5 ms each in one Python process.  Forks, so call it
    BEFORE the process has touched the GPU (torch.cuda.is_available() initialises it); a process that already has builds
    them in line.  The result is what the sequential loop gives (every MDP is a pure function of its config and seed)."""
    import os
    seeds = list(seeds)
    n = workers if workers is not None else min(os.cpu_count() or 1, 64)
    gpu_touched = False
    try:
        import torch
        gpu_touched = torch.cuda.is_initialized()
    except Exception:
        pass
    if n <= 1 or len(seeds) < 512 or gpu_touched:
        return _build_chunk((config, seeds))
    import contextlib
    import io
    import multiprocessing as mp
    step = max(64, -(-len(seeds) // (4 * n)))
    chunks = [(config, seeds[k:k + step]) for k in range(0, len(seeds), step)]
    with mp.get_context("fork").Pool(n) as pool, contextlib.redirect_stdout(io.StringIO()):
        parts = pool.map(_build_chunk, chunks)
    return [m for part in parts for m in part]
