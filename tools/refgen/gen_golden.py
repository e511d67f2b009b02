#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *reference* RLToyEnv.

THIS CONTAINER ONLY.  Needs /root/reference (read-only) and the gymnasium API
stand-in under tools/refgen/gymnasium_standin (gymnasium itself is not
installed here; see DESIGN.md "Oracle pinning").  Nothing in here travels to
the GPU box except the .npz / .json it writes under tests/golden/.

    python tools/refgen/gen_golden.py            # regenerate everything

What is recorded per case (arrays are [E, T, ...]; E = env instances, one per
seed; T = steps):
  * the config dict (JSON) and the seed of every instance
  * everything __init__ generated: transition matrix, rewardable sequences,
    terminal states, init-state distribution, seed_dict
  * the PCG64 state of every generator step()/reset() draws from, captured
    right after construction
  * the action fed at every step, whether reset() was called after the step
  * per step: obs, reward (float64), done; internal curr_state / derivatives
"""
import contextlib
import io
import json
import logging
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden"))


def make_env(config):
    from mdp_playground.envs.rl_toy_env import RLToyEnv

    cfg = dict(config)
    if isinstance(cfg.get("seed"), dict):
        cfg["seed"] = dict(cfg["seed"])
    cfg["log_level"] = logging.CRITICAL
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = RLToyEnv(**cfg)
    return env


def pcg_state(gen):
    st = gen.bit_generator.state
    assert st["bit_generator"] == "PCG64"
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array(
        [s & m, s >> 64, inc & m, inc >> 64, st["has_uint32"], st["uinteger"]],
        dtype=np.uint64,
    )


BASE_D = dict(state_space_type="discrete", action_space_type="discrete",
              state_space_size=8, action_space_size=8)
CFG1 = dict(BASE_D, delay=0)
CFG2 = dict(BASE_D, delay=4, sequence_length=3)
CFG2N = dict(CFG2, transition_noise=0.2, reward_noise=0.3, reward_scale=2.5,
             reward_shift=-1.75, term_state_reward=-0.5, reward_every_n_steps=1)
BASE_C = dict(state_space_type="continuous", state_space_dim=12,
              relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
              target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10,
              action_space_max=1, inertia=1, make_denser=True,
              reward_function="move_to_a_point")
CFG3 = dict(BASE_C, transition_dynamics_order=1, time_unit=1)
CFG5 = dict(BASE_C, transition_dynamics_order=2, time_unit=0.1,
            transition_noise=0.05, reward_noise=0.05)
CFG4 = dict(CFG1, image_representations=True, image_width=84, image_height=84,
            image_transforms="shift,rotate", image_sh_quant=1, image_ro_quant=1)

CASES = {
    # --- discrete ------------------------------------------------------
    "d_cfg1": dict(config=CFG1, seeds=list(range(8)), T=150, reset="on_done"),
    "d_cfg2": dict(config=CFG2, seeds=list(range(16)), T=300, reset="on_done"),
    "d_cfg2_noreset": dict(config=CFG2, seeds=list(range(4)), T=60, reset="never"),
    "d_cfg2_noise": dict(config=CFG2N, seeds=list(range(8)), T=300, reset="mixed"),
    "d_s16_l2_d1": dict(
        config=dict(BASE_D, state_space_size=16, action_space_size=16, delay=1,
                    sequence_length=2, reward_every_n_steps=2, reward_density=0.1,
                    transition_noise=0.1, reward_scale=0.5),
        seeds=list(range(4)), T=200, reset="on_done"),
    "d_l1_d0_pnoise": dict(
        config=dict(BASE_D, delay=0, sequence_length=1, transition_noise=0.5,
                    terminal_state_density=0.125),
        seeds=list(range(4)), T=200, reset="mixed"),
    "d_l4_repeats": dict(
        config=dict(BASE_D, delay=2, sequence_length=4, repeats_in_sequences=True,
                    reward_density=0.05, reward_noise=1.0),
        seeds=list(range(4)), T=200, reset="on_done"),
    "d_rdist": dict(
        config=dict(BASE_D, delay=1, sequence_length=2, reward_dist=[0.25, 1.0]),
        seeds=list(range(4)), T=200, reset="on_done"),
    "d_diam2": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=12, action_space_size=6, diameter=2, delay=0,
                    sequence_length=3, terminal_state_density=0.34),
        seeds=list(range(4)), T=200, reset="on_done"),
    # (round 6: state spaces beyond 255 states -- 16-bit table entries and history fields, mdpp_discrete_wide.hip)
    "d_s300_noise": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=300, action_space_size=300, delay=1, sequence_length=1,
                    reward_density=0.25, terminal_state_density=0.05, transition_noise=0.1, reward_noise=0.2, reward_scale=1.5),
        seeds=list(range(3)), T=120, reset="on_done"),
    "d_s300_diam50_l2": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=300, action_space_size=6, diameter=50, delay=2,
                    sequence_length=2, terminal_state_density=0.34),
        seeds=list(range(3)), T=300, reset="on_done"),
    # (round 6: sequence_length beyond 7 -- a history of sixteen byte fields, mdpp_discrete_long.hip)
    "d_l9_repeats": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=4, action_space_size=4, delay=2, sequence_length=9, repeats_in_sequences=True,
                    reward_density=0.3, terminal_state_density=0.25, reward_noise=0.1),
        seeds=list(range(3)), T=200, reset="on_done"),
    "d_l8_s5": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=5, action_space_size=5, delay=0, sequence_length=8, repeats_in_sequences=True,
                    reward_density=0.5, terminal_state_density=0.2, reward_dist=[0.01, 1]),
        seeds=list(range(2)), T=300, reset="on_done"),
    "d_notmax": dict(
        config=dict(BASE_D, delay=0, sequence_length=2, maximally_connected=False),
        seeds=list(range(4)), T=100, reset="on_done"),
    # the reference's own passing tests use seed *dicts*
    "d_kat_everyn": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=8, action_space_size=8, reward_density=0.25,
                    make_denser=False, terminal_state_density=0.25,
                    maximally_connected=True, repeats_in_sequences=False, delay=0,
                    sequence_length=3, reward_scale=1.0, generate_random_mdp=True,
                    seed={"env": 0, "relevant_state_space": 8,
                          "relevant_action_space": 8}),
        seeds=[None], T=6, reset="never", actions=[6, 2, 2, 4, 4, 6]),
    # --- use_custom_mdp with P and R given as matrices (lists here, arrays when handed to the env)
    "d_custom_pr": dict(     # the env of the reference's test_discrete_custom_P_R (:1990-2036)
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=8, action_space_size=5, terminal_state_density=0.25,
                    repeats_in_sequences=False, delay=1, reward_scale=2.0, use_custom_mdp=True,
                    transition_function=np.random.default_rng(0).integers(8, size=(8, 5)).tolist(),
                    reward_function=np.random.default_rng(1).integers(4, size=(8, 5)).tolist(),
                    init_state_dist=[1 / 8 for _ in range(8)]),
        seeds=list(range(4)), T=200, reset="on_done"),
    "d_custom_noise": dict(  # float rewards, own terminal states and rho_0, both noises, every-n 2
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=12, action_space_size=6, delay=3, reward_every_n_steps=2,
                    reward_scale=1.5, reward_shift=-0.25, term_state_reward=-0.5,
                    transition_noise=0.2, reward_noise=0.1, use_custom_mdp=True,
                    transition_function=np.random.default_rng(5).integers(12, size=(12, 6)).tolist(),
                    reward_function=np.round(np.random.default_rng(6).normal(size=(12, 6)), 3).tolist(),
                    terminal_states=[3, 7],
                    init_state_dist=(np.arange(1, 13) * (np.arange(12) % 4 != 3) / 54.0).tolist()),
        seeds=list(range(4)), T=200, reset="mixed"),
    # --- discrete with an irrelevant sub-space (Tuple spaces; second table, second P-noise stream)
    "d_irr_plain": dict(     # the config of the reference's test_discrete_irr_features (:1729-1774)
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=[8, 10], action_space_size=[8, 10],
                    irrelevant_features=True, reward_density=0.25, make_denser=True,
                    terminal_state_density=0.25, maximally_connected=True,
                    repeats_in_sequences=False, delay=1, sequence_length=1,
                    reward_scale=1.0, generate_random_mdp=True),
        seeds=list(range(4)), T=120, reset="on_done"),
    "d_irr_noise": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=[8, 5], action_space_size=[8, 5],
                    irrelevant_features=True, delay=2, sequence_length=2,
                    transition_noise=0.3, reward_noise=0.2, reward_scale=1.5,
                    reward_shift=0.25, term_state_reward=-1.0),
        seeds=list(range(6)), T=200, reset="mixed"),
    "d_irr_notmax": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=[6, 12], action_space_size=[6, 4],
                    irrelevant_features=True, delay=0, sequence_length=3,
                    maximally_connected=False, transition_noise=0.1),
        seeds=list(range(4)), T=150, reset="on_done"),
    # --- image observations of Tuple spaces: one polygon image per sub-space, side by side --------
    "i_irr": dict(
        config=dict(state_space_type="discrete", action_space_type="discrete",
                    state_space_size=[8, 11], action_space_size=[8, 11], irrelevant_features=True,
                    delay=0, image_representations=True, image_width=84, image_height=84,
                    image_transforms="shift,rotate,flip", image_sh_quant=2, image_ro_quant=3),
        seeds=list(range(3)), T=24, reset="on_done"),
    # --- grid envs: 2-D, delay 0, sequence_length 1 (anything else raises
    # inside the reference's reward_function) --------------------------------------------------
    "g_dense": dict(      # the env of the reference's test_grid_env (:1057-1110)
        config=dict(state_space_type="grid", grid_shape=(8, 8), delay=0, sequence_length=1,
                    reward_function="move_to_a_point", make_denser=True, target_point=[5, 5],
                    reward_scale=3.0, terminal_states=[[5, 5], [2, 3], [2, 4], [3, 3], [3, 4]],
                    term_state_reward=-0.25),
        seeds=list(range(6)), T=150, reset="on_done", bad_action_every=23),
    "g_noise_sparse": dict(
        config=dict(state_space_type="grid", grid_shape=(5, 7), reward_function="move_to_a_point",
                    make_denser=False, target_point=[2, 3], transition_noise=0.3, reward_noise=0.2,
                    reward_every_n_steps=2, reward_shift=0.5, term_state_reward=1.5),
        seeds=list(range(6)), T=200, reset="mixed", bad_action_every=31),
    "g_irr": dict(
        config=dict(state_space_type="grid", grid_shape=(6, 9), reward_function="move_to_a_point",
                    make_denser=True, target_point=[1, 7], irrelevant_features=True,
                    transition_noise=0.2, reward_scale=0.5),
        seeds=list(range(4)), T=200, reset="on_done"),
    # --- continuous ------------------------------------------------------
    "c_cfg3": dict(config=CFG3, seeds=list(range(8)), T=250, reset="on_done",
                   bad_action_every=37),
    "c_cfg5": dict(config=CFG5, seeds=list(range(8)), T=250, reset="on_done",
                   bad_action_every=41),
    "c_order3_delay3": dict(
        config=dict(BASE_C, transition_dynamics_order=3, time_unit=0.3, inertia=2.0,
                    transition_noise=0.02, delay=3, reward_scale=2.0,
                    reward_shift=0.5, action_loss_weight=0.1),
        seeds=list(range(4)), T=250, reset="on_done", bad_action_every=50),
    "c_sparse_term": dict(
        config=dict(state_space_type="continuous", state_space_dim=2,
                    transition_dynamics_order=1, inertia=1.0, time_unit=1.0,
                    state_space_max=5, action_space_max=1, make_denser=False,
                    target_point=[1.0, -1.0], target_radius=1.5,
                    terminal_states=[[-3.0, 3.0], [3.0, 3.0]], term_state_edge=2.0,
                    term_state_reward=-1.0, reward_scale=2.0,
                    reward_function="move_to_a_point"),
        seeds=list(range(8)), T=200, reset="on_done"),
    # twelve terminal hypercubes (round 5: the device limit was 8; the reference has none, rl_toy_env.py:891-956) in a
    # 3-D space with one irrelevant dimension, order 2, delay 1
    "c_many_boxes": dict(
        config=dict(state_space_type="continuous", state_space_dim=3, relevant_indices=[0, 2], irrelevant_features=True,
                    transition_dynamics_order=2, inertia=1.0, time_unit=0.5,
                    state_space_max=6, action_space_max=1, make_denser=True, delay=1,
                    target_point=[0.5, -0.5], target_radius=0.4,
                    terminal_states=[[-5.0, -5.0], [-5.0, 0.0], [-5.0, 5.0], [0.0, -5.0], [0.0, 5.0], [5.0, -5.0],
                                     [5.0, 0.0], [5.0, 5.0], [-2.5, 2.5], [2.5, -2.5], [-2.5, -2.5], [2.5, 2.5]],
                    term_state_edge=1.5, term_state_reward=-2.0, reward_scale=1.5,
                    reward_function="move_to_a_point"),
        seeds=list(range(8)), T=200, reset="on_done"),
    "c_small_radius_hit": dict(
        config=dict(state_space_type="continuous", state_space_dim=2,
                    transition_dynamics_order=1, inertia=1.0, time_unit=1.0,
                    state_space_max=2, action_space_max=1, make_denser=True,
                    target_point=[0.0, 0.0], target_radius=0.8, reward_noise=0.1,
                    reward_every_n_steps=2, reward_function="move_to_a_point"),
        seeds=list(range(8)), T=200, reset="mixed"),
    # --- the DEFAULT target_point (:652-654): float64 zeros of length state_space_dim -- usable when every dimension is
    # relevant -- which promotes distances, the target latch and a dense reward to float64 (np.float64 throughout)
    "c_default_target": dict(
        config=dict(state_space_type="continuous", state_space_dim=4, transition_dynamics_order=2,
                    inertia=1.0, time_unit=0.5, state_space_max=3, action_space_max=1, make_denser=True,
                    target_radius=0.6, delay=2, reward_noise=0.1, reward_scale=1.5, reward_shift=-0.25,
                    term_state_reward=2.0, action_loss_weight=0.05, reward_function="move_to_a_point"),
        seeds=list(range(4)), T=200, reset="mixed", bad_action_every=31),
    "c_default_target_sparse": dict(
        config=dict(state_space_type="continuous", state_space_dim=2, transition_dynamics_order=1,
                    inertia=1.0, time_unit=1.0, state_space_max=2, action_space_max=1, make_denser=False,
                    target_radius=0.9, reward_every_n_steps=2, transition_noise=0.05, reward_noise=0.2,
                    action_loss_weight=0.1, reward_function="move_to_a_point"),
        seeds=list(range(4)), T=200, reset="mixed"),
    # --- the env object's per-episode noise statistics (logged at every reset(), :2231-2247) recorded per step ------
    "d_stats": dict(config=dict(CFG2N, reward_every_n_steps=2), seeds=list(range(4)), T=160, reset="mixed", stats=True),
    "g_stats": dict(
        config=dict(state_space_type="grid", grid_shape=(5, 7), reward_function="move_to_a_point",
                    make_denser=True, target_point=[2, 3], transition_noise=0.3, reward_noise=0.2,
                    reward_scale=1.5, term_state_reward=1.5),
        seeds=list(range(4)), T=160, reset="mixed", stats=True),
    "c_stats": dict(
        config=dict(BASE_C, state_space_dim=4, irrelevant_features=False, relevant_indices=[0, 1, 2, 3],
                    target_radius=1.5, state_space_max=4, transition_dynamics_order=2, time_unit=0.5,
                    transition_noise=0.05, reward_noise=0.1, delay=1, reward_every_n_steps=2, reward_scale=2.0),
        seeds=list(range(4)), T=160, reset="mixed", stats=True),
    # --- reward_function move_along_a_line: random actions, then the same
    # action repeated (the rewards of a straight walk are LAPACK-rounding-sized) -------------------
    "c_line_4d": dict(       # the env of the reference's test_continuous_dynamics_move_along_a_line
        config=dict(state_space_type="continuous", state_space_dim=4,
                    transition_dynamics_order=1, inertia=1, time_unit=1, delay=0,
                    sequence_length=10, reward_scale=1.0, action_space_max=1,
                    reward_function="move_along_a_line"),
        seeds=list(range(4)), T=140, reset="mixed", straight_window=14),
    "c_line_irr": dict(
        config=dict(state_space_type="continuous", state_space_dim=6, irrelevant_features=True,
                    relevant_indices=[1, 3, 4], transition_dynamics_order=2, inertia=1.0,
                    time_unit=0.5, state_space_max=8, action_space_max=1, delay=2,
                    sequence_length=5, reward_noise=0.05, reward_scale=2.0, reward_shift=0.5,
                    terminal_states=[[3.0, 3.0, 3.0]], term_state_edge=9.0, term_state_reward=-1.0,
                    reward_function="move_along_a_line"),
        seeds=list(range(4)), T=160, reset="mixed", straight_window=9, bad_action_every=29),
    "c_line_6of8": dict(     # six relevant dimensions (rows of 8 in the line history, the 8 x 8 scatter matrix)
        config=dict(state_space_type="continuous", state_space_dim=8, irrelevant_features=True,
                    relevant_indices=[0, 2, 3, 4, 6, 7], transition_dynamics_order=1, inertia=1.0,
                    time_unit=1.0, state_space_max=6, action_space_max=1, delay=1,
                    sequence_length=7, reward_scale=1.5,
                    reward_function="move_along_a_line"),
        seeds=list(range(3)), T=120, reset="mixed", straight_window=11),
    "c_line_12of14": dict(   # twelve relevant dimensions: past the register-resident fits (c_line_reward_big, round 6)
        config=dict(state_space_type="continuous", state_space_dim=14, irrelevant_features=True,
                    relevant_indices=[0, 1, 2, 4, 5, 6, 7, 8, 10, 11, 12, 13], transition_dynamics_order=1, inertia=1.0,
                    time_unit=1.0, state_space_max=6, action_space_max=1, delay=1,
                    sequence_length=7, reward_scale=1.5,
                    reward_function="move_along_a_line"),
        seeds=list(range(3)), T=120, reset="mixed", straight_window=11),
    # --- continuous + ImageContinuous observations: RGB pictures, and the
    # reference's quirk that every step takes the clip-and-zero-derivatives branch -----------------
    "ci_2d": dict(
        config=dict(state_space_type="continuous", state_space_dim=2, transition_dynamics_order=2,
                    inertia=1.0, time_unit=1.0, state_space_max=5, action_space_max=1,
                    make_denser=True, target_point=[1.0, -1.0], target_radius=0.5,
                    terminal_states=[[-3.0, 3.0], [3.0, 3.0]], term_state_edge=2.0,
                    reward_function="move_to_a_point", image_representations=True,
                    image_width=100, image_height=100),
        seeds=[0, 1, 2], T=40, reset="on_done", bad_action_every=17),
    "ci_4d_noise": dict(
        config=dict(state_space_type="continuous", state_space_dim=4, relevant_indices=[0, 1],
                    transition_dynamics_order=1, inertia=1.0, time_unit=0.5, state_space_max=3,
                    action_space_max=1, make_denser=False, target_point=[-1.0, 2.0],
                    target_radius=0.6, transition_noise=0.3, reward_noise=0.1,
                    reward_function="move_to_a_point", image_representations=True,
                    image_width=64, image_height=48),
        seeds=[0, 1], T=40, reset="mixed"),
    # (round 6: more drawn rectangles than the 8 the picture kernel's argument block used to hold: 14 cubes, overlapping ones too)
    "ci_14_boxes": dict(
        config=dict(state_space_type="continuous", state_space_dim=2, transition_dynamics_order=1,
                    inertia=1.0, time_unit=1.0, state_space_max=8, action_space_max=1,
                    make_denser=True, target_point=[0.5, -0.5], target_radius=0.4,
                    terminal_states=[[-6.0, -6.0], [-6.0, 0.0], [-6.0, 6.0], [0.0, -6.0], [0.0, 6.0], [6.0, -6.0], [6.0, 0.0],
                                     [6.0, 6.0], [-3.0, 3.0], [3.0, -3.0], [-3.0, -3.0], [3.0, 3.0], [3.5, 3.5], [-7.5, 7.5]],
                    term_state_edge=1.5, term_state_reward=-1.0,
                    reward_function="move_to_a_point", image_representations=True,
                    image_width=84, image_height=84),
        seeds=[0, 1, 2], T=48, reset="on_done"),
    # (round 6: a pixel count that is not a multiple of 16 -- the byte-store form of k_imagec_obs; 4 state dimensions: two pictures)
    "ci_50x50_4d": dict(
        config=dict(state_space_type="continuous", state_space_dim=4, transition_dynamics_order=1,
                    inertia=1.0, time_unit=1.0, state_space_max=4, action_space_max=1, relevant_indices=[0, 1],
                    make_denser=False, target_point=[1.0, 1.0], target_radius=0.8, terminal_states=[[-2.0, -2.0]], term_state_edge=1.0,
                    reward_function="move_to_a_point", image_representations=True,
                    image_width=50, image_height=50),
        seeds=[0, 1], T=40, reset="on_done"),
    # --- grid + ImageContinuous observations (grid lines, terminal cells drawn as rectangles) ----
    "gi_45x35": dict(
        config=dict(state_space_type="grid", grid_shape=(5, 7), reward_function="move_to_a_point",
                    make_denser=False, target_point=[3, 4], terminal_states=[[1, 1]], transition_noise=0.2,
                    image_representations=True, image_width=45, image_height=35),
        seeds=[0, 1], T=40, reset="on_done"),
    "gi_12_cells": dict(
        config=dict(state_space_type="grid", grid_shape=(9, 7), reward_function="move_to_a_point",
                    make_denser=True, target_point=[4, 3],
                    terminal_states=[[0, 0], [0, 6], [8, 0], [8, 6], [2, 2], [2, 4], [6, 2], [6, 4], [4, 0], [4, 6], [0, 3], [8, 3]],
                    image_representations=True, image_width=72, image_height=56),
        seeds=[0, 1], T=40, reset="on_done"),
    "gi_8x8": dict(
        config=dict(state_space_type="grid", grid_shape=(8, 8), reward_function="move_to_a_point",
                    make_denser=True, target_point=[5, 5], reward_scale=3.0,
                    terminal_states=[[2, 3], [7, 7], [0, 4]], term_state_reward=-0.25,
                    image_representations=True, image_width=96, image_height=80),
        seeds=[0, 1, 2], T=40, reset="on_done", bad_action_every=13),
    "gi_irr_noise": dict(
        config=dict(state_space_type="grid", grid_shape=(4, 6), reward_function="move_to_a_point",
                    make_denser=False, target_point=[1, 2], irrelevant_features=True,
                    transition_noise=0.25, reward_noise=0.1, terminal_states=[[0, 0]],
                    image_representations=True, image_width=64, image_height=64),
        seeds=[0, 1], T=40, reset="mixed"),
    # --- discrete + image observations -------------------------------------
    "i_cfg4": dict(config=CFG4, seeds=[0, 1], T=24, reset="on_done"),
    "i_100_all": dict(
        config=dict(CFG1, image_representations=True, image_width=100,
                    image_height=100, image_transforms="shift,scale,rotate,flip",
                    image_sh_quant=2, image_ro_quant=5,
                    image_scale_range=(0.5, 1.5)),
        seeds=[0, 1], T=24, reset="on_done"),
    # the reference's own image sweeps (experiments/a3c_image_representations.py: 100 x 100, image_scale_range (0.5, 2),
    # arm "shift,scale,rotate,flip", default quantisation): radii 9 ... 41 -- the wide-template renderer
    "i_100_sweep": dict(
        config=dict(CFG1, image_representations=True, image_width=100,
                    image_height=100, image_transforms="shift,scale,rotate,flip",
                    image_scale_range=(0.5, 2)),
        seeds=[0, 1, 2], T=32, reset="on_done"),
    # a quantised shift that leaves the draw's own range: (v // q) * q rounds towards -inf, so with q = 5 at 84 x 84 (mw = 22, v in
    # -21 .. 21) the centre moves by -25 and the polygon is CLIPPED by the picture's edge (3 px) before the rotation samples it
    # (the reference's own sh_quant sweeps have it: 100 x 100, q = 4 / 8 / 16 -> -32 against mw = 30)
    "i_shq5_rot": dict(
        config=dict(CFG1, image_representations=True, image_width=84,
                    image_height=84, image_transforms="shift,rotate,flip",
                    image_sh_quant=5, image_ro_quant=3),
        seeds=[0, 1, 2, 3], T=48, reset="on_done"),
    "i_none": dict(
        config=dict(CFG1, image_representations=True, image_width=84,
                    image_height=84),
        seeds=[3], T=10, reset="on_done"),
}


def jsonable(cfg):
    def conv(v):
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, (np.floating,)):
            return float(v)
        if isinstance(v, (list, tuple)):
            return [conv(x) for x in v]
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        return v
    return conv(cfg)


def run_case(name, case):
    base = case["config"]
    kind = base["state_space_type"]
    image = bool(base.get("image_representations", False))
    T = case["T"]
    rec = {k: [] for k in ("obs", "reward", "done", "action", "reset_after",
                           "reset_obs", "curr_state")}
    if case.get("stats"):
        # the env object's per-episode noise statistics after every step (before any reset() of that step):
        # total_abs_noise_in_reward_episode, total_reward_episode, total_noisy_transitions_episode,
        # total_transitions_episode, [total_abs_noise_in_transition_episode per dimension] (:2360-2369)
        rec["stats"] = []
    tables = {k: [] for k in ("P", "terminal_states", "init_dist", "rew_keys",
                              "rew_vals", "rng_env", "rng_space", "rng_image",
                              "init_obs", "init_state", "seed_dict", "sd",
                              "P_irr", "init_dist_irr", "rng_space_irr", "rng_action",
                              "rew_matrix")}
    seed_names = ["env", "relevant_state_space", "relevant_action_space",
                  "irrelevant_state_space", "irrelevant_action_space", "state_space",
                  "action_space", "image_representations"]
    for e, seed in enumerate(case["seeds"]):
        cfg = dict(base)
        if seed is not None:
            cfg["seed"] = seed
        if cfg.get("use_custom_mdp"):
            for k in ("transition_function", "reward_function", "init_state_dist"):
                cfg[k] = np.array(cfg[k])
        env = make_env(cfg)
        arng = np.random.default_rng(1000003 * (e + 1) + 17)  # action/reset policy rng
        sdict = env.seed_dict
        tables["seed_dict"].append(
            np.array([sdict.get(k, -1) if sdict.get(k, -1) is not None else -1
                      for k in seed_names], dtype=np.int64))
        tables["rng_env"].append(pcg_state(env._np_random))
        if kind == "discrete":
            S = int(env.state_space_size[0])
            L = env.sequence_length
            tables["P"].append(np.array(env.transition_matrix.tolist(), dtype=np.int64))
            tables["terminal_states"].append(
                np.array(env.config["terminal_states"], dtype=np.int64))
            tables["init_dist"].append(
                np.array(env.config["relevant_init_state_dist"], dtype=np.float64))
            keys, vals = [], []
            for seq, v in getattr(env, "rewardable_sequences", {}).items():
                if len(seq) == L:
                    keys.append(list(seq))
                    vals.append(float(v))
            tables["rew_keys"].append(np.array(keys, dtype=np.int64).reshape(-1, L))
            tables["rew_vals"].append(np.array(vals, dtype=np.float64))
            tables["rng_space"].append(pcg_state(env.observation_spaces[0].np_random))
            if env.use_custom_mdp:
                tables["rew_matrix"].append(np.array(env.reward_matrix, dtype=np.float64))
            if env.irrelevant_features:
                tables["P_irr"].append(np.array(
                    env.config["transition_function_irrelevant"].tolist(), dtype=np.int64))
                tables["init_dist_irr"].append(
                    np.array(env.config["irrelevant_init_state_dist"], dtype=np.float64))
                tables["rng_space_irr"].append(pcg_state(env.observation_spaces[1].np_random))
            if image:
                tables["rng_image"].append(pcg_state(env.observation_space.np_random))
        else:
            tables["rng_space"].append(pcg_state(env.feature_space.np_random))
            if kind == "grid":
                tables["rng_action"].append(pcg_state(env.action_space.np_random))
        co = env.curr_obs  # NB: __init__ stores reset()'s (obs, info) tuple here
        tables["init_obs"].append(np.array(co[0] if isinstance(co, tuple) else co))
        tables["init_state"].append(np.array(env.curr_state))
        r = {k: [] for k in rec}
        sdrec = []
        for t in range(T):
            if kind == "discrete":
                if env.irrelevant_features:
                    a = [int(arng.integers(env.action_space_size[0])),
                         int(arng.integers(env.action_space_size[1]))]
                elif "actions" in case:
                    a = int(case["actions"][t])
                else:
                    a = int(arng.integers(env.action_space_size[0]))
                act = a
                r["action"].append(a)
            elif kind == "grid":
                G = len(env.grid_shape)
                a = [0] * G
                u = arng.random()
                if u < 0.45:                           # head for the target: episodes end often
                    cs, tp = [int(x) for x in env.curr_state], env.target_point
                    i = int(arng.integers(2))
                    if cs[i] == tp[i]:
                        i = 1 - i
                    a[i] = int(np.sign(tp[i] - cs[i]))
                elif u < 0.95:
                    a[int(arng.integers(G))] = int(arng.integers(3)) - 1
                bae = case.get("bad_action_every")
                if bae and t % bae == bae - 1:        # not in the action space: applied as a noop
                    a = [1] * G if (t // bae) % 2 else [2] + [0] * (G - 1)
                act = a
                r["action"].append(list(a))
            else:
                D = env.state_space_dim
                amax = env.action_space_max
                a = arng.uniform(-amax, amax, D).astype(np.float32)
                sw = case.get("straight_window")
                if sw and (t // sw) % 2 == 1 and t % sw != 0:   # second window: repeat the last action
                    a = r["action"][-1].copy()
                bae = case.get("bad_action_every")
                if bae and t % bae == bae - 1:
                    a[int(arng.integers(D))] = np.float32(amax * 1.5)
                act = a
                r["action"].append(a.copy())
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                obs, rew, done, trunc, info = env.step(act)
            r["obs"].append(np.array(obs).copy())
            r["reward"].append(np.float64(rew))
            r["done"].append(bool(done))
            r["curr_state"].append(np.array(env.curr_state).copy())
            if "stats" in rec:
                row = [float(env.total_abs_noise_in_reward_episode), float(env.total_reward_episode),
                       float(env.total_noisy_transitions_episode), float(env.total_transitions_episode)]
                if kind == "continuous":
                    row += [float(x) for x in env.total_abs_noise_in_transition_episode]
                r["stats"].append(np.array(row, dtype=np.float64))
            if kind == "continuous":
                sdrec.append(np.stack([np.array(x, dtype=np.float32)
                                       for x in env.state_derivatives]))
            mode = case["reset"]
            do_reset = (mode == "on_done" and done) or (
                mode == "mixed" and ((done and arng.random() < 0.6)
                                     or arng.random() < 0.02))
            r["reset_after"].append(bool(do_reset))
            if do_reset:
                ro, _ = env.reset()
                r["reset_obs"].append(np.array(ro).copy())
            else:
                r["reset_obs"].append(np.zeros_like(np.array(obs)))
        for k in rec:
            rec[k].append(np.stack(r[k]))
        if kind == "continuous":
            tables["sd"].append(np.stack(sdrec))
        env.close()

    out = {}
    for k, v in rec.items():
        out[k] = np.stack(v)
    ragged = ("rew_keys", "rew_vals", "terminal_states")
    for k, v in tables.items():
        if not v:
            continue
        if k in ragged:
            for e, arr in enumerate(v):
                out[f"{k}_{e}"] = arr
        else:
            out[k] = np.stack(v)
    out["reward"] = out["reward"].astype(np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    n_done = int(out["done"].sum())
    n_reset = int(out["reset_after"].sum())
    print(f"{name}: E={len(case['seeds'])} T={T} dones={n_done} resets={n_reset}")


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    meta = {}
    for name, case in CASES.items():
        meta[name] = jsonable({k: v for k, v in case.items()})
        if only and name not in only:
            continue
        run_case(name, case)
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    import PIL
    import scipy
    with open(os.path.join(OUT, "VERSIONS.txt"), "w") as f:
        f.write(f"python {sys.version.split()[0]}\nnumpy {np.__version__}\n"
                f"scipy {scipy.__version__}\npillow {PIL.__version__}\n"
                "gymnasium: API stand-in (tools/refgen/gymnasium_standin)\n"
                "reference: automl/mdp-playground @ 2025-09-26\n")


if __name__ == "__main__":
    main()
