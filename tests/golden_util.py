"""Helpers shared by the parity tests: load golden fixtures (tests/golden/*.npz,
made by tools/refgen/gen_golden.py from the reference) and build oracle instances
from the tables recorded in them."""
import json
import os

import numpy as np

from oracle import oracle as ora

GOLDEN = os.environ.get("MDPP_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")    # (the override: scratch
#                                     sets made by tools/refgen/gen_sweep.py, run through the same tests before any of them is committed)
# tests/golden_sweep/: the env configurations of the reference's OWN experiment sweeps (/root/reference/experiments/*.py, every
# RLToy-v0 file: a star over its var_env_configs, duplicates merged -- tools/refgen/gen_sweep.py), recorded from the reference
# like every other golden; names ?_x<nnn> fall into the lists below by their prefix, so every golden-driven test runs on them too
SWEEP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_sweep")

with open(os.path.join(GOLDEN, "cases.json")) as _f:
    CASES = json.load(_f)
_DIR = {k: GOLDEN for k in CASES}
if os.path.exists(os.path.join(SWEEP, "cases.json")) and not os.environ.get("MDPP_GOLDEN_DIR"):
    with open(os.path.join(SWEEP, "cases.json")) as _f:
        for _k, _v in json.load(_f).items():
            CASES[_k] = _v
            _DIR[_k] = SWEEP

IRRELEVANT = sorted(k for k in CASES if k.startswith("d_irr"))      # Tuple spaces: pairs of states / actions
DISCRETE = sorted(k for k in CASES if k.startswith("d_") and k not in IRRELEVANT)
CONTINUOUS = sorted(k for k in CASES if k.startswith("c_"))
IMAGE = sorted(k for k in CASES if k.startswith("i_"))
GRID = sorted(k for k in CASES if k.startswith("g_"))
IMAGE_GRID = sorted(k for k in CASES if k.startswith("gi_"))     # grid envs with ImageContinuous observations
IMAGE_CONT = sorted(k for k in CASES if k.startswith("ci_"))    # continuous envs with ImageContinuous observations


def load(name):
    return np.load(os.path.join(_DIR.get(name, GOLDEN), name + ".npz"))


def case_config(name, e=0):
    cfg = dict(CASES[name]["config"])
    seed = CASES[name]["seeds"][e]
    if seed is not None:
        cfg["seed"] = seed
    return cfg


def discrete_params(cfg):
    """Scalar parameters exactly as rl_toy_env.py:342-566 defaults them."""
    L = cfg.get("sequence_length", 1)
    return dict(
        L=L, delay=cfg.get("delay", 0),
        every_n=cfg.get("reward_every_n_steps", L),
        transition_noise=cfg.get("transition_noise"),
        reward_noise=cfg.get("reward_noise"),
        reward_scale=cfg.get("reward_scale", 1.0),
        reward_shift=cfg.get("reward_shift", 0.0),
        term_state_reward=cfg.get("term_state_reward", 0.0))


def discrete_oracle_from_golden(name, g, e):
    cfg = CASES[name]["config"]
    p = discrete_params(cfg)
    P = g["P"][e]
    S, A = P.shape
    rt = ora.rtable_from_sequences(S, p["L"], g[f"rew_keys_{e}"], g[f"rew_vals_{e}"])
    o = ora.DiscreteOracle(S, A, p["L"], p["delay"], p["every_n"], P, rt,
                           g[f"terminal_states_{e}"], g["init_dist"][e],
                           p["transition_noise"], p["reward_noise"], p["reward_scale"],
                           p["reward_shift"], p["term_state_reward"])
    if "rew_matrix" in g.files:                      # use_custom_mdp with matrices
        o.set_reward_matrix(g["rew_matrix"][e])
    if "P_irr" in g.files:
        o.set_irrelevant(g["P_irr"][e], g["init_dist_irr"][e])
    return o


def continuous_params(cfg):
    D = cfg["state_space_dim"]
    rel = list(cfg.get("relevant_indices", range(D)))
    boxes_lo = boxes_hi = None
    if "terminal_states" in cfg:
        ts = np.array(cfg["terminal_states"], dtype=np.float64)
        edge = cfg["term_state_edge"]
        boxes_lo = (ts - edge / 2).astype(np.float32)
        boxes_hi = (ts + edge / 2).astype(np.float32)
    return dict(
        D=D, relevant_indices=rel, order=cfg.get("transition_dynamics_order", 1),
        inertia=cfg.get("inertia", 1.0), time_unit=cfg.get("time_unit", 1.0),
        state_space_max=cfg.get("state_space_max", np.inf),
        action_space_max=cfg.get("action_space_max", np.inf),
        target_point=cfg.get("target_point", [0.0] * len(rel)), target_radius=cfg.get("target_radius", 0.05),
        make_denser=cfg.get("make_denser", True),
        action_loss_weight=cfg.get("action_loss_weight", 0.0),
        transition_noise=cfg.get("transition_noise"), reward_noise=cfg.get("reward_noise"),
        delay=cfg.get("delay", 0), every_n=cfg.get("reward_every_n_steps", 1),
        reward_scale=cfg.get("reward_scale", 1.0), reward_shift=cfg.get("reward_shift", 0.0),
        term_state_reward=cfg.get("term_state_reward", 0.0),
        box_lo=boxes_lo, box_hi=boxes_hi)


def continuous_oracle_from_golden(name):
    cfg = CASES[name]["config"]
    o = ora.ContinuousOracle(**continuous_params(cfg))
    if cfg.get("reward_function") == "move_along_a_line":
        o.set_line_reward(cfg.get("sequence_length", 1), cfg.get("delay", 0))
    elif "target_point" not in cfg:
        o.set_target64()
    return o


def grid_params(cfg):
    """Scalar parameters of a grid env as rl_toy_env.py:342-566 defaults them."""
    shape = list(cfg["grid_shape"]) * (2 if cfg.get("irrelevant_features") else 1)   # :604-608
    return dict(grid_shape=shape, target_point=list(cfg["target_point"]), make_denser=cfg["make_denser"],
                transition_noise=cfg.get("transition_noise"), reward_noise=cfg.get("reward_noise"),
                every_n=cfg.get("reward_every_n_steps", 1), reward_scale=cfg.get("reward_scale", 1.0),
                reward_shift=cfg.get("reward_shift", 0.0), term_state_reward=cfg.get("term_state_reward", 0.0))


def grid_oracle_from_golden(name):
    return ora.GridOracle(**grid_params(CASES[name]["config"]))
