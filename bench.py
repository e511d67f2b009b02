#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched RLToyEnv.step() hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (starts its N ranks itself, as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): discrete 8 states x 8 actions, reward_delay 4,
sequence_length 3, 65 536 env instances per GPU sharing one MDP, uniform random actions
(synthetic, pre-generated on the device), same-step autoreset, numpy-exact PCG64 streams.

ONE BENCH STEP = one pass of the hot path over one batch = ONE fused launch (mdpp_step_n) of
--fuse (512, the same for every --gpus) env steps of every env instance of every rank, i.e.
`--steps 20 --warmup 5` is 5 + 20 launches of 512 x 65 536 env steps per GPU.  `value` stays in
env-steps/s (bench steps x fuse x envs x ranks / time), `ms_per_step` is per bench step (launch).

`value` is the SAME experiment for every --gpus, one rank included: each launch is followed by ONE RCCL
all-gather that assembles the current global observation tensor on every rank (env ids are sharded contiguously,
weak scaling; the gather runs on a side stream and overlaps the next launch) -- a plain one-GPU run makes a
one-rank RCCL group of its own.  `value_none` is the same launches without any collective, `value_last_row` the
leg with it (= `value`); `collective_legs` adds a gather of every observation of the rollout ([K, N_local, ...]).

Reads are honest: the timed launches cycle through >= 4 distinct action tensors, >= 512 MiB together (above the
256 MiB Infinity Cache); `roofline.frac` is that leg, `roofline.frac_replayed` the same launches replaying one
tensor (what rounds 1-2 reported).

`workloads`: after the cfg2 leg the other BASELINE configs run under the same clock on one GPU -- cfg3, cfg4, cfg5
(numpy-exact and Philox streams) and cfg2 on Philox streams, >= 10 fused launches each, HIP events on the launch
stream: {name: {env_steps_per_s, launch_us, frac, kernel, traffic}}.
The single-launch-per-step path (mdpp_step) is reported beside it.

Rank 0 prints ONE JSON line: the driver contract plus `roofline` (HIP events around the timed
launches; PMC traffic from two child rocprofv3 passes of the same launches; the measured copy / write ceilings of
this device beside the 8 TB/s spec peak) and `cpu_baseline` (a pure-Python restatement of the reference step(),
baseline/py_step.py; the C port of the oracle is reported as `cpu_baseline_port`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the host driver only supports dmabuf IPC (RCCL, CUDA-tensor sharing): must be in the environment before the HIP runtime
# reads it, i.e. before anything touches the GPU (setting it after torch.cuda.set_device had no effect)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import statistics  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0    # what the guide measures for a float4 copy kernel on MI355X (MI355X_MICROARCH.md)

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                 config=dict(state_space_type="discrete", action_space_type="discrete",
                             state_space_size=8, action_space_size=8, delay=4,
                             sequence_length=3, seed=0)),
    # cfg2 with both noises on (every step draws from two PCG64 streams)
    "cfg2_noise": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                       config=dict(state_space_type="discrete", action_space_type="discrete",
                                   state_space_size=8, action_space_size=8, delay=4,
                                   sequence_length=3, transition_noise=0.1, reward_noise=0.1,
                                   seed=0)),
    # BASELINE.json configs[2]
    "cfg3": dict(kind="continuous", envs=65536, alg_bytes_fused=102, alg_bytes_step=206,
                 config=dict(state_space_type="continuous", state_space_dim=12,
                             relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
                             target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10,
                             action_space_max=1, transition_dynamics_order=1, inertia=1,
                             time_unit=1, make_denser=True, reward_function="move_to_a_point",
                             seed=0)),
    # BASELINE.json configs[3]: 84x84 polygon images, shift + rotate, 8 192 envs
    "cfg4": dict(kind="discrete", envs=8192, alg_bytes_fused=7082, alg_bytes_step=7082,
                 config=dict(state_space_type="discrete", action_space_type="discrete",
                             state_space_size=8, action_space_size=8, delay=0,
                             image_representations=True, image_width=84, image_height=84,
                             image_transforms="shift,rotate", image_sh_quant=1, image_ro_quant=1,
                             seed=0)),
    # the reference's own image sweeps (/root/reference/experiments/a3c_image_representations.py: 100 x 100 pictures,
    # image_scale_range (0.5, 2), arm "shift,scale,rotate,flip"): radii 10 ... 40 -- templates past the 64-byte LDS columns of
    # k_image_obs_fast (k_image_obs_wide since round 5; the general renderer before)
    "img100_all": dict(kind="discrete", envs=8192, alg_bytes_fused=10026, alg_bytes_step=10026, fuse_max=64,
                       config=dict(state_space_type="discrete", action_space_type="discrete",
                                   state_space_size=8, action_space_size=8, delay=0,
                                   image_representations=True, image_width=100, image_height=100,
                                   image_transforms="shift,scale,rotate,flip", image_scale_range=(0.5, 2),
                                   seed=0)),
    # ... the arms without the scale transform ("shift", "rotate", "flip", "none": 27 of its 38 image configurations): 64-byte templates
    "img100_shift": dict(kind="discrete", envs=8192, alg_bytes_fused=10026, alg_bytes_step=10026, fuse_max=64,
                         config=dict(state_space_type="discrete", action_space_type="discrete",
                                     state_space_size=8, action_space_size=8, delay=0,
                                     image_representations=True, image_width=100, image_height=100,
                                     image_transforms="shift", seed=0)),
    # BASELINE.json configs[4] (per-GPU shard of the 524 288-env job)
    "cfg5": dict(kind="continuous", envs=65536, alg_bytes_fused=102, alg_bytes_step=350,
                 config=dict(state_space_type="continuous", state_space_dim=12,
                             relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
                             target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10,
                             action_space_max=1, transition_dynamics_order=2, inertia=1,
                             time_unit=0.1, transition_noise=0.05, reward_noise=0.05,
                             make_denser=True, reward_function="move_to_a_point", seed=0)),
    # the discrete shapes k_discrete_rollout_lean does not take -- what the reference's own sweeps use:
    # /root/reference/experiments/dqn_delay_50_states.py (S = A = 50, sequence_length 1, delay in {0, 1, 2, 4, 8}),
    # rainbow_reward_dist.py (S = A = 24, reward_dist: non-unit rewards), and one MDP PER ENV (seeds=[...], what every golden
    # uses: tables gathered from HBM / L2 instead of staged in LDS)
    "d_s50_delay4": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                         config=dict(state_space_type="discrete", action_space_type="discrete", state_space_size=50,
                                     action_space_size=50, delay=4, sequence_length=1, reward_density=0.25,
                                     terminal_state_density=0.25, seed=0)),
    "d_s24_rdist": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                        config=dict(state_space_type="discrete", action_space_type="discrete", state_space_size=24,
                                    action_space_size=24, delay=0, sequence_length=1, reward_density=0.25,
                                    terminal_state_density=0.25, reward_dist=[0.01, 1], seed=0)),
    # the commonest discrete shape of the reference's experiment files (a3c_del, dqn_del, rainbow_del, ... at delay 0): every one of
    # them passes reward_noise: 0 and transition_noise: 0, and the reference DRAWS rng.normal(0, 0) per step for a zero sigma
    # (rl_toy_env.py:398-403, :1982) -- stream-exact parity keeps the draw, so this shape runs the kernels WITH reward noise
    "d_s8_rn0": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                     config=dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
                                 delay=0, sequence_length=1, reward_density=0.25, terminal_state_density=0.25, make_denser=False,
                                 transition_noise=0, reward_noise=0, reward_scale=1.0, completely_connected=True,
                                 generate_random_mdp=True, repeats_in_sequences=False, seed=0)),
    # ... the same with 50 states (dqn_*_50_states: the quiet kernel with the reward-noise draw inside the recurrence)
    "d_s50_rn0": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42,
                      config=dict(state_space_type="discrete", action_space_type="discrete", state_space_size=50, action_space_size=50,
                                  delay=0, sequence_length=1, reward_density=0.25, terminal_state_density=0.25, make_denser=False,
                                  transition_noise=0, reward_noise=0, reward_scale=1.0, completely_connected=True,
                                  generate_random_mdp=True, repeats_in_sequences=False, seed=0)),
    # ... and its commonest continuous shape (ddpg / td3 / sac_move_to_a_point_*: two dimensions, order 1, BOTH noise keys at 0 --
    # three normals drawn per step for nothing)
    "c_d2_n0": dict(kind="continuous", envs=65536, alg_bytes_fused=22, alg_bytes_step=22 + 16 + 8,
                    config=dict(state_space_type="continuous", action_space_type="continuous", state_space_dim=2, action_space_dim=2,
                                transition_dynamics_order=1, inertia=1, time_unit=1.0, state_space_max=10, action_space_max=1,
                                target_point=[0, 0], target_radius=0.5, make_denser=True, reward_function="move_to_a_point",
                                action_loss_weight=0.01, delay=0, reward_scale=1.0, transition_noise=0, reward_noise=0, seed=0)),
    # (65 536 envs since round 6: 8 192 lanes are 128 waves on 1 024 SIMDs -- that leg timed an empty chip; the 8 192-env figure
    #  stays beside it as cfg2_per_env_8k)
    # ... the same env without the noise keys (not a default leg: the base D = 2 kernel, for profiles)
    "c_d2": dict(kind="continuous", envs=65536, alg_bytes_fused=22, alg_bytes_step=22 + 16 + 8,
                 config=dict(state_space_type="continuous", action_space_type="continuous", state_space_dim=2, action_space_dim=2,
                             transition_dynamics_order=1, inertia=1, time_unit=1.0, state_space_max=10, action_space_max=1,
                             target_point=[0, 0], target_radius=0.5, make_denser=True, reward_function="move_to_a_point",
                             action_loss_weight=0.01, delay=0, reward_scale=1.0, seed=0)),
    "cfg2_per_env": dict(kind="discrete", envs=65536, alg_bytes_fused=18, alg_bytes_step=42, per_env_mdps=True,
                         config=dict(state_space_type="discrete", action_space_type="discrete",
                                     state_space_size=8, action_space_size=8, delay=4, sequence_length=3)),
    "cfg2_per_env_8k": dict(kind="discrete", envs=8192, alg_bytes_fused=18, alg_bytes_step=42, per_env_mdps=True,
                            config=dict(state_space_type="discrete", action_space_type="discrete",
                                        state_space_size=8, action_space_size=8, delay=4, sequence_length=3)),
    # SURVEY.md §8f rank 2 (not a BASELINE config): the reference's test_grid_env shape, 65 536 envs
    "grid": dict(kind="grid", envs=65536, alg_bytes_fused=30, alg_bytes_step=62,
                 config=dict(state_space_type="grid", grid_shape=(8, 8), reward_function="move_to_a_point",
                             make_denser=True, target_point=[5, 5], reward_scale=3.0,
                             term_state_reward=-0.25, seed=0)),
    # SURVEY.md §8f rank 3 (not a BASELINE config): continuous env with ImageContinuous observations,
    # 100x100 RGB (the reference's defaults), 8 192 envs; 30 000 B written per env step
    "img_cont": dict(kind="continuous", envs=8192, alg_bytes_fused=30000 + 8 + 6, alg_bytes_step=30000 + 8 + 6 + 48,
                     fuse_max=32,      # 246 MB of pictures per step: 32 steps = 7.9 GB per rollout buffer
                     config=dict(state_space_type="continuous", state_space_dim=2, transition_dynamics_order=1,
                                 inertia=1.0, time_unit=1.0, state_space_max=5, action_space_max=1,
                                 make_denser=True, target_point=[1.0, -1.0], target_radius=0.5,
                                 terminal_states=[[-3.0, 3.0], [3.0, 3.0]], term_state_edge=2.0,
                                 reward_function="move_to_a_point", image_representations=True,
                                 image_width=100, image_height=100, seed=0)),
    # §8f rank 2, last item (not a BASELINE config): reward_function move_along_a_line, the env of the
    # reference's test_continuous_dynamics_move_along_a_line; the per-step line fit is f64 arithmetic
    "line": dict(kind="continuous", envs=65536, alg_bytes_fused=38, alg_bytes_step=38 + 16 + 24 + 8 + 160,
                 config=dict(state_space_type="continuous", state_space_dim=4, transition_dynamics_order=1,
                             inertia=1, time_unit=1, delay=0, sequence_length=10, reward_scale=1.0,
                             action_space_max=1, state_space_max=100, reward_function="move_along_a_line",
                             seed=0)),
    # the irrelevant-sub-space variant of cfg2's MDP size (Tuple spaces), also §8f rank 2
    "cfg2_irr": dict(kind="discrete", envs=65536, alg_bytes_fused=30, alg_bytes_step=62,
                     config=dict(state_space_type="discrete", action_space_type="discrete",
                                 state_space_size=[8, 8], action_space_size=[8, 8],
                                 irrelevant_features=True, delay=4, sequence_length=3, seed=0)),
}


def make_actions(wl, K, N, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if wl["kind"] == "discrete":
        A = wl["config"]["action_space_size"]
        if isinstance(A, (list, tuple)):       # irrelevant_features: (relevant, irrelevant) pairs
            return torch.stack([torch.randint(0, a, (K, N), generator=g, device=device, dtype=torch.int32)
                                for a in A], dim=2).contiguous()
        return torch.randint(0, A, (K, N), generator=g, device=device, dtype=torch.int32)
    if wl["kind"] == "grid":                    # one +-1 (or a noop) in a random dimension
        G = len(wl["config"]["grid_shape"]) * (2 if wl["config"].get("irrelevant_features") else 1)
        which = torch.randint(0, G, (K, N, 1), generator=g, device=device)
        val = torch.randint(-1, 2, (K, N, 1), generator=g, device=device, dtype=torch.int32)
        return torch.zeros((K, N, G), dtype=torch.int32, device=device).scatter_(2, which, val)
    D = wl["config"]["state_space_dim"]
    amax = wl["config"]["action_space_max"]
    return (torch.rand((K, N, D), generator=g, device=device, dtype=torch.float32) * 2 - 1) * amax


def cpu_baseline(wl, seconds=6.0):
    """The oracle (a scalar C port of the reference step(), oracle/mdpp_oracle.c) timed on ONE
    host core over a bounded sample of the same workload: 64 env instances stepped with random
    actions and reset-on-done until ~`seconds` of CPU time have been spent."""
    from mdp_playground_amd import mdp as mdp_mod
    from oracle import oracle as ora
    m = mdp_mod.build_mdp(wl["config"])
    n_envs, chunk = 64, 20000
    rng = np.random.default_rng(12345)
    envs = []
    for i in range(n_envs):
        if m.kind == "grid":
            o = ora.GridOracle(m.grid_shape, m.target_point, m.make_denser, m.transition_noise,
                               m.reward_noise, m.reward_every_n_steps, m.reward_scale, m.reward_shift,
                               m.term_state_reward)
            o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator((m.seed_dict["env"] or 0) + i)),
                      mdp_mod.pcg64_words(mdp_mod.new_generator(m.seed_dict["state_space"] + i)),
                      mdp_mod.pcg64_words(mdp_mod.new_generator(m.seed_dict["action_space"] + i)))
            o.reset()
            envs.append(o)
            continue
        if m.kind == "discrete":
            o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps,
                                   m.P, m.reward_table(), m.terminal_states, m.init_dist,
                                   m.transition_noise, m.reward_noise, m.reward_scale,
                                   m.reward_shift, m.term_state_reward)
            sp = mdp_mod.new_generator(m.space_seeds[0] + i)
            if m.irrelevant:
                o.set_irrelevant(m.P_irr, m.init_dist_irr)
                o.set_rng_irr(mdp_mod.pcg64_words(mdp_mod.new_generator(m.space_seeds[1] + i)))
        else:
            o = ora.ContinuousOracle(m.D, m.relevant_indices, m.order, m.inertia, m.time_unit,
                                     m.state_space_max, m.action_space_max, m.target_point,
                                     m.target_radius, m.make_denser, m.action_loss_weight,
                                     m.transition_noise, m.reward_noise, m.delay,
                                     m.reward_every_n_steps, m.reward_scale, m.reward_shift,
                                     m.term_state_reward, m.box_lo, m.box_hi)
            if m.reward_function == "move_along_a_line":
                o.set_line_reward(m.sequence_length, m.delay)
            sp = mdp_mod.new_generator(m.seed_dict["state_space"] + i)
        o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator((m.seed_dict["env"] or 0) + i)),
                  mdp_mod.pcg64_words(sp))
        o.reset()
        envs.append(o)
    if m.kind == "grid":
        G = len(m.grid_shape)
        acts = np.zeros((chunk, G), np.int32)
        acts[np.arange(chunk), rng.integers(0, G, size=chunk)] = rng.integers(-1, 2, size=chunk)
    elif m.kind == "discrete" and m.irrelevant:
        acts = np.stack([rng.integers(0, m.A, size=chunk), rng.integers(0, m.A_irr, size=chunk)], axis=1).astype(np.int32)
    elif m.kind == "discrete":
        acts = rng.integers(0, m.A, size=chunk).astype(np.int32)
    else:
        acts = rng.uniform(-m.action_space_max, m.action_space_max, size=(chunk, m.D)).astype(np.float32)
    steps, spent = 0, 0.0
    while spent < seconds:
        for o in envs:
            t0 = time.perf_counter()
            o.rollout(acts, None)
            spent += time.perf_counter() - t0
            steps += chunk
            if spent >= seconds:
                break
    return {"value": steps / spent, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} env-steps of the same workload ({n_envs} instances, reset on done) "
                      f"through oracle/mdpp_oracle.c on 1 host core in {spent:.1f} s; "
                      f"host has {os.cpu_count()} cores"}


def _cpu_worker(args):
    wl_name, seconds, wid = args
    wl = WORKLOADS[wl_name]
    from mdp_playground_amd import mdp as mdp_mod
    from oracle import oracle as ora
    m = mdp_mod.build_mdp(wl["config"])
    o = ora.DiscreteOracle(m.S, m.A, m.sequence_length, m.delay, m.reward_every_n_steps, m.P,
                           m.reward_table(), m.terminal_states, m.init_dist, m.transition_noise,
                           m.reward_noise, m.reward_scale, m.reward_shift, m.term_state_reward)
    o.set_rng(mdp_mod.pcg64_words(mdp_mod.new_generator(1000 + wid)), m.space_rng_words)
    o.reset()
    acts = np.random.default_rng(wid).integers(0, m.A, size=20000).astype(np.int32)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        o.rollout(acts, None)
        steps += len(acts)
    return steps, time.perf_counter() - t0


def cpu_baseline_all_cores(wl_name, seconds=2.0):
    """Same C port, one process per host core (discrete workloads), embarrassingly parallel."""
    import contextlib
    import io
    import multiprocessing as mp
    n = os.cpu_count() or 1
    with mp.get_context("fork").Pool(n) as pool, contextlib.redirect_stdout(io.StringIO()):
        res = pool.map(_cpu_worker, [(wl_name, seconds, w) for w in range(n)])
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": total / wall, "unit": "env-steps/s", "cores": n, "kind": "port",
            "sample": f"{total} env-steps, one oracle process per core for {wall:.1f} s"}


def hbm_ceilings(device, nbytes=1 << 30, reps=10):
    """What the memory system of THIS device gives plain streaming kernels, measured live (SURVEY.md
    §8d "report against both"), two ways: the library's own 16-bytes-per-lane one-shot kernels (mdpp_probe_hbm: the
    float4 copy MI355X_MICROARCH.md quotes at 6.29 TB/s, a fill, a read; copy and fill with plain and with non-temporal
    stores, the FASTER form is the ceiling -- `copy_GBps`, `write_GBps`, `read_GBps`, the forms in `forms_GBps`) and torch's
    copy_ / fill_ kernels (`torch_copy_GBps`, `torch_write_GBps`), HIP events around `reps` launches over `nbytes`.  A
    ceiling is the best a plain kernel reaches: `write_GBps` = max over all fill forms incl. torch's, `copy_GBps` likewise
    (rounds 3-4 ran the probe on a persistent grid, 25 % under a one-shot fill, and the rollout kernel beat it)."""
    import ctypes
    from mdp_playground_amd import _capi
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    out = {}
    lib = _capi.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    forms = {}
    for name, mode, moved in (("copy", 0, 2 * nbytes), ("write", 1, nbytes), ("read", 2, nbytes), ("copy_nt", 3, 2 * nbytes),
                              ("write_nt", 4, nbytes)):
        ms = ctypes.c_float(0.0)
        rc = lib.mdpp_probe_hbm(mode, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), nbytes, reps, stream,
                                ctypes.byref(ms))
        if rc == 0 and ms.value > 0:
            forms[name] = moved * reps / (ms.value * 1e-3) / 1e9
    for name in ("copy", "write", "read"):
        best = max([v for k, v in forms.items() if k.split("_")[0] == name], default=None)
        if best is not None:
            out[name + "_GBps"] = best
    out["forms_GBps"] = forms
    torch_out = {}
    for name, fn, moved in (("copy", lambda: dst.copy_(src), 2 * nbytes), ("write", lambda: dst.fill_(7), nbytes)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        torch_out[name] = moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    out["torch_copy_GBps"], out["torch_write_GBps"] = torch_out["copy"], torch_out["write"]
    out["copy_GBps"] = max(out.get("copy_GBps", 0.0), torch_out["copy"])
    out["write_GBps"] = max(out.get("write_GBps", 0.0), torch_out["write"])
    return out


def action_rotation(wl, F, N, device, seed, min_total=512 << 20, min_n=4, max_n=32):
    """Distinct pre-generated action tensors, cycled through the launches.  Together they are larger than the
    256 MiB Infinity Cache (>= 512 MiB, at least 4 tensors), so the action reads of a launch come from HBM and
    not from what the previous replay of the same tensor left in the cache."""
    first = make_actions(wl, F, N, device, seed)
    nbytes = first.numel() * first.element_size()
    n = min(max_n, max(min_n, -(-min_total // nbytes)))
    return [first] + [make_actions(wl, F, N, device, seed + 1000 * j) for j in range(1, n)]


# the other BASELINE.json configs (and cfg2 on the RNG north_star names), timed in the same run after the cfg2 leg
EXTRA_LEGS = (("cfg2", "philox"), ("cfg3", "numpy"), ("cfg4", "numpy"), ("cfg5", "numpy"), ("cfg5", "philox"),
              ("cfg2_noise", "numpy"), ("cfg2_noise", "philox"),   # (+ cfg2 with both noises, reference-exact streams and the north_star RNG)
              ("d_s50_delay4", "numpy"), ("d_s24_rdist", "numpy"), ("cfg2_per_env", "numpy"),   # (+ the discrete shapes beyond the lean kernel)
              ("img100_all", "numpy"),                                                           # (+ the reference's own image sweep shape)
              ("d_s8_rn0", "numpy"), ("d_s8_rn0", "philox"), ("d_s50_rn0", "numpy"),                                    # (+ its commonest discrete shape: noise keys with sigma 0)
              ("c_d2_n0", "numpy"), ("c_d2_n0", "philox"))                                       # (+ ... and continuous shape: D = 2, both noise keys 0)


def leg_name(workload, rng):
    return workload if rng == "numpy" else f"{workload}_{rng}"


PER_ENV_CACHE = "/tmp/mdpp_bench_per_env_%d_%d.pkl"       # (pid of the bench process, envs)


def prebuild_per_env(wl, N, cache_pid=None):
    """The N different MDPs of a `per_env_mdps` workload, built by a pool of forked workers BEFORE this process touches the GPU
    (mdp.build_many) and pickled for the PMC child, which runs under rocprofv3 and must not fork."""
    import pickle
    from mdp_playground_amd import mdp as mdp_mod
    mdps = mdp_mod.build_many(wl["config"], range(N))
    wl["_mdps"] = mdps
    if cache_pid is not None:
        try:
            with open(PER_ENV_CACHE % (cache_pid, N), "wb") as f:
                pickle.dump(mdps, f, protocol=pickle.HIGHEST_PROTOCOL)
        except OSError:
            pass
    return mdps


def make_env(wl, N, device, rng, **kw):
    """The workload's batched env (`per_env_mdps`: env i is built from seed i -- N different MDPs, tables per env)."""
    from mdp_playground_amd import RLToyVectorEnv
    if wl.get("per_env_mdps"):
        mdps = wl.get("_mdps")
        if mdps is None or len(mdps) != N:
            mdps = None
            cache = PER_ENV_CACHE % (int(os.environ.get("MDPP_BENCH_PID", "0")), N)
            if os.path.exists(cache):       # (the PMC child: what the bench process built)
                import pickle
                with open(cache, "rb") as f:
                    mdps = pickle.load(f)
        return RLToyVectorEnv(seeds=list(range(N)), mdps=mdps, device=device, rng=rng, autoreset="same_step", **kw, **wl["config"])
    return RLToyVectorEnv(num_envs=N, device=device, rng=rng, autoreset="same_step", **kw, **wl["config"])


def workload_leg(name, rng, device, fuse, launches, warmup, seed=12345, repeats=5):
    """One more workload under the same clock: `warmup` untimed + `launches` timed fused rollouts (HIP events on
    the launch stream around the timed ones, rotating action tensors), everything resident in HBM."""
    from mdp_playground_amd import RLToyVectorEnv
    wl = WORKLOADS[name]
    N = wl["envs"]
    F = max(1, min(fuse, wl.get("fuse_max", fuse)))
    env = make_env(wl, N, device, rng)
    acts = action_rotation(wl, F, N, device, seed)
    out = env.alloc_rollout(F)
    for it in range(max(warmup, 1)):
        env.rollout(acts[it % len(acts)], out)
    torch.cuda.synchronize(device)
    us, walls, n = [], [], warmup
    for _ in range(max(repeats, 1)):            # R repeats of `launches` launches: the median is reported
        env.timer_begin()
        t0 = time.perf_counter()
        for it in range(launches):
            env.rollout(acts[(n + it) % len(acts)], out)
        ms = env.timer_end()
        torch.cuda.synchronize(device)
        walls.append(time.perf_counter() - t0)
        us.append(ms * 1e3 / launches)
        n += launches
    bad = int((env.status() != 0).sum())
    kname = env.rollout_kernel_name(F)
    single = None
    if name in ("cfg3", "cfg4", "cfg5", "d_s50_delay4", "img100_all"):    #  the one-launch-per-step API and a replayed HIP graph of 64 such steps
        try:
            single = single_step_leg(env, wl, acts[0], N, device, n1=200, reps=5)
        except Exception as e:                  # a reported extra, never fatal
            single = {"error": repr(e)}
    env.close()
    per_launch_s = statistics.median(us) * 1e-6
    wall = statistics.median(walls)
    alg = wl["alg_bytes_fused"] * N * F
    achieved = alg / per_launch_s / 1e9
    return {"env_steps_per_s": N * F * launches / wall, "launch_us": per_launch_s * 1e6, "launches": launches,
            "repeats": len(us), "launch_us_runs": {"min": min(us), "median": statistics.median(us), "max": max(us)},
            "frac": achieved / HBM_PEAK_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
            "achieved_GBps": achieved, "kernel": kname, "rng": rng,
            "envs": N, "fuse": F, "alg_bytes_per_env_step": wl["alg_bytes_fused"], "alg_bytes_per_launch": alg,
            "action_tensors": len(acts), "action_bytes_rotated": len(acts) * acts[0].numel() * acts[0].element_size(),
            "envs_with_status_bits": bad, "traffic": None, "traffic_source": None, "single_step": single}


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a
    CHILD process (this process has not touched the GPU and never will), relay rank 0's JSON line and
    the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def init_collective(device, world, no_collective, backend="nccl"):
    """The process group of the run: RCCL (backend "nccl").  Launched by torch.distributed.run it joins that
    job; a plain one-GPU run makes a ONE-rank group of its own, so that N = 1 times the same code path
    (launch + all-gather on a side stream) as N > 1 and a 1 -> 8 curve compares like with like."""
    if no_collective:
        return None, "disabled (--no-collective)"
    import torch.distributed as dist
    if backend == "gloo":               # (tests: several ranks on ONE GPU, the rank logic of the multi-rank path)
        dist.init_process_group(backend="gloo")
        return dist, None
    try:
        # the collectives' stream at high priority: its hardware queue is then not one the launch stream can share
        # (streams of one priority are spread over a few hardware queues; a shared queue serialises gather and launch)
        opts = None
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            pass
        if "RANK" in os.environ:
            dist.init_process_group(backend="nccl", device_id=device, pg_options=opts)
        else:
            if world != 1:
                raise RuntimeError("no launcher")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    device_id=device, pg_options=opts)
        return dist, None
    except Exception as e:              # a one-rank run still has its `none` leg
        if world > 1:
            raise
        return None, f"unavailable: {e!r}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed bench steps; ONE bench step = one fused launch of "
                    "--fuse env steps of every env instance of every rank")
    ap.add_argument("--warmup", type=int, default=5, help="untimed bench steps (launches) before the timed ones")
    ap.add_argument("--fuse", type=int, default=512, help="env steps per fused launch (mdpp_step_n), the same for every --gpus")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="env instances per GPU")
    ap.add_argument("--rng", default="numpy", choices=["numpy", "philox"])
    ap.add_argument("--disable", default="", help="comma-separated mdpp_set_options switches (include/mdpp.h MDPP_OPT_*, "
                    "e.g. NO_HELPER): take specialised kernels out of the dispatch, for A/B timings")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baselines on every host core (extras `cpu_baseline_all_cores`, "
                    "`cpu_baseline_port_all_cores`; + ~10 s)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live PMC traffic measurement (two child rocprofv3 runs)")
    ap.add_argument("--cpu-baselines-only", default=None, metavar="WORKLOAD", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", nargs=2, default=None, metavar=("SPECS_JSON", "OUT"), help=argparse.SUPPRESS)
    ap.add_argument("--peer-copy", action="store_true", help="also time the hipIpc peer-copy gather (mdpp_peer_*) beside the RCCL leg; "
                    "runs last, never `value`")
    ap.add_argument("--no-collective", action="store_true", help="one-GPU runs: no RCCL group, `value` = the leg without a collective")
    ap.add_argument("--no-workloads", action="store_true", help="skip the legs of the other BASELINE configs (`workloads`)")
    ap.add_argument("--workload-steps", type=int, default=10, help="timed launches of each `workloads` leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only for tests that run several ranks on one GPU (RCCL is the product path)")
    ap.add_argument("--repeats", type=int, default=5, help="every timed leg is repeated this many times; the MEDIAN repeat is "
                    "reported (`value`, `ms_per_step`, `roofline.launch_us`), all repeats in `*_runs`")
    ap.add_argument("--only-leg", default=None, choices=["rotating", "replayed"],
                    help="profiling: time only this form of the main workload's launches (no collective, no other legs), so that a "
                    "rocprofv3 --kernel-trace --stats of the run holds ONE leg per kernel (profiles/r04_kernel_stats_<leg>.csv)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="the FULL record (every leg, every repeat, prose) goes to this file; stdout gets the compact contract line only")
    ap.add_argument("--print-detail", action="store_true", help="also print the full record, prefixed BENCH_DETAIL, before the contract line")
    ap.add_argument("--full-gather-steps", type=int, default=4,
                    help="bench steps of the [K, N_local, ...] all-gather leg (0 = skip)")
    args = ap.parse_args()

    if args.cpu_baselines_only:       # a child of rank 0 of a multi-rank run (fresh process, never touches the GPU)
        cpu_py, cpu_py_all, cpu_all = cpu_baselines_forked(args.cpu_baselines_only, args.cpu_all_cores)
        print("CPU_BASELINES " + json.dumps({"cpu_py": cpu_py, "cpu_py_all": cpu_py_all, "cpu_all": cpu_all}), flush=True)
        return
    if args.pmc_child:                # a child of rank 0 (never touches the GPU itself: it starts the rocprofv3 passes), beside the CPU baselines
        res = live_traffic_all([tuple(x) for x in json.loads(args.pmc_child[0])])
        with open(args.pmc_child[1], "w") as f:
            json.dump(res, f)
        return
    # preflight: RCCL refuses two ranks on one device -- say so in one line instead of hanging in the rendezvous
    # (torch.cuda.device_count() does not initialise the GPU runtime on this image)
    if args.backend == "nccl" and args.gpus > 1 and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} with backend nccl (RCCL) needs {args.gpus} visible devices, "
                         f"torch.cuda.device_count() = {torch.cuda.device_count()} (several ranks on one device: --backend gloo)")
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CPU baselines first: they fork one worker per core, which must happen before this process has
    # initialised the GPU runtime
    wl = WORKLOADS[args.workload]
    N = args.envs or wl["envs"]
    F = max(1, min(args.fuse, wl.get("fuse_max", args.fuse)))
    extra = []
    if args.only_leg is not None:       # a profiling run of one leg of the main kernel: nothing else on the device
        args.no_workloads = args.no_cpu_baseline = args.no_pmc = args.no_single_step = args.no_collective = True
    if world == 1 and args.workload == "cfg2" and args.rng == "numpy" and not args.no_workloads and not args.disable \
            and args.envs is None:
        extra = list(EXTRA_LEGS)
    phases, _t_last = {}, [time.perf_counter()]

    def mark(name):                     # wall seconds of each phase of this process (detail record: `timing_s`)
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - _t_last[0], 2)
        _t_last[0] = now

    cpu_py = cpu_py_all = cpu_all = None
    # one rank: now, before this process initialises the GPU (the baselines fork one worker per core).  Several ranks: at the
    # END, from a fresh child of rank 0, after the process group is gone -- not while ranks 1..N-1 wait in the rendezvous for
    # a rank 0 that is busy on every core for half a minute
    # The PMC passes (child rocprofv3 runs of tools/pmc_workloads.py: counters, not times) run BESIDE the CPU baselines, started by
    # a child of their own -- this process has not touched the GPU yet and the two do not share anything that is timed
    # (the GPU is idle while the host cores are timed; BENCH_r05: 89 s of driver time for a 2.4 ms timed region).
    pmc, pmc_proc, pmc_out = None, None, None
    per_env_cached = []
    for w, r in ([(args.workload, args.rng)] + extra) if rank == 0 else []:        # N different MDPs per env: built by forked workers, now
        if WORKLOADS[w].get("per_env_mdps") and "_mdps" not in WORKLOADS[w]:
            n_ = N if w == args.workload else WORKLOADS[w]["envs"]
            prebuild_per_env(WORKLOADS[w], n_, os.getpid())
            per_env_cached.append(PER_ENV_CACHE % (os.getpid(), n_))
    mark("per_env_mdps")
    if rank == 0 and world == 1 and not args.no_pmc and not _pmc_blocked():
        import subprocess
        import tempfile
        specs = [(args.workload, args.rng, N, F)] + [
            (w, r, WORKLOADS[w]["envs"], max(1, min(args.fuse, WORKLOADS[w].get("fuse_max", args.fuse)))) for w, r in extra]
        fd, pmc_out = tempfile.mkstemp(prefix="mdpp_pmc_", suffix=".json", dir="/tmp")
        os.close(fd)
        pmc_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--pmc-child", json.dumps(specs), pmc_out],
                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True,
                                    env=dict(os.environ, MDPP_BENCH_PID=str(os.getpid())))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_py, cpu_py_all, cpu_all = cpu_baselines_forked(args.workload, args.cpu_all_cores)
    mark("cpu_baselines_forked")
    if pmc_proc is not None:
        try:
            pmc_proc.wait(timeout=400)
            with open(pmc_out) as f:
                pmc = json.load(f)
        except Exception:               # (a reported extra: the line then carries the committed traffic record)
            import signal
            try:
                os.killpg(pmc_proc.pid, signal.SIGKILL)
            except OSError:
                pass
            pmc = None
        finally:
            try:
                os.unlink(pmc_out)
            except OSError:
                pass
    mark("pmc_passes_wait")
    for f_ in per_env_cached:
        try:
            os.unlink(f_)
        except OSError:
            pass
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist, no_coll_why = init_collective(device, world, args.no_collective, args.backend)
    backend_used = args.backend if dist is not None else None

    from mdp_playground_amd import RLToyVectorEnv
    from mdp_playground_amd.dist import ObsGatherer

    env = RLToyVectorEnv(num_envs=N, device=device, env_id_offset=rank * N, rng=args.rng,
                         autoreset="same_step", **wl["config"])
    if args.disable:
        env.set_kernel_options(*args.disable.split(","))
    acts = action_rotation(wl, F, N, device, 12345 + rank)
    NA = len(acts)
    mark("init")
    # output buffers rotate, so a gather can trail a launch; four with a collective: launch k + 1 must not wait for
    # the gather of launch k - 2, which slips in between two launches (a rollout holds every CU)
    outs = [env.alloc_rollout(F) for _ in range(4 if dist is not None else 2)]
    NB = len(outs)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    launched = [0]
    host_enqueue = [0.0]

    def run(steps, gathers, rotate=True):
        """`steps` fused launches; with `gathers`, each launch is followed by ONE all-gather of the tensor gathers[j]
        was built on, started asynchronously (ObsGatherer.start): the backend's stream waits for the launch that fills
        it and runs the collective beside the next launch; nothing is inserted into the launch stream."""
        works = [None] * NB
        for it in range(steps):
            j = it % NB
            # the gather that read this buffer NB launches ago: asked on the host first -- when it has completed
            # (the usual case) no wait goes into the launch stream, and launches stay back to back
            if gathers is not None and works[j] is not None and not works[j].is_completed():
                works[j].wait()
            env.rollout(acts[launched[0] % NA] if rotate else acts[0], outs[j])
            launched[0] += 1
            if gathers is not None:
                works[j] = gathers[j].start()
        if gathers is not None:
            for w in works:
                if w is not None:
                    getattr(w, "finish", w.wait)()  # (the current stream waits for the collective's stream / the peers' shards)

    def timed(steps, gathers, events=False, rotate=True):
        """EXACTLY `steps` launches between barrier + synchronize on both sides; max over ranks."""
        barrier()
        if events:
            env.timer_begin()
        t0 = time.perf_counter()
        run(steps, gathers, rotate)
        host_enqueue[0] = time.perf_counter() - t0      # the host's share: everything is asynchronous up to here
        kernel_ms = env.timer_end() if events else None
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        barrier()
        if dist is not None and world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, kernel_ms

    # (Every leg runs on the default stream.  A probe that picked "the quickest of five candidate launch streams" was tried in
    # round 3 and taken out again: the candidates measured the same with the collective, and a picked stream that happened to
    # share a hardware queue with an image handle's side stream serialised cfg4's prepare and render stages -- 7.4 or 8.4 ms per
    # launch from run to run.)
    R = max(1, args.repeats)

    def timed_reps(steps, gathers, events=False, rotate=True):
        """R back-to-back repeats of timed(): [(elapsed_s, kernel_ms)]; the MEDIAN repeat is what a leg reports
        (one 2.9 ms sample decided `value` in round 3; box noise is +-5 %)."""
        return [timed(steps, gathers, events, rotate) for _ in range(R)]

    def leg_record(reps, steps):
        els = [e for e, _ in reps]
        med = statistics.median(els)
        return med, {"elapsed_s": med, "env_steps_per_s": world * N * F * steps / med,
                     "elapsed_s_runs": els, "env_steps_per_s_runs": [world * N * F * steps / e for e in els]}

    # ---- leg "none": no collective.  Its HIP-event time (launch stream) is the roofline leg: every launch reads a
    # different action tensor (NA of them, > the Infinity Cache together)
    legs = {}
    kernel_us_runs = replay_us_runs = None
    el_none = None
    if args.only_leg in (None, "rotating"):
        run(max(args.warmup, 1), None)
        reps = timed_reps(args.steps, None, events=True)
        el_none, legs["none"] = leg_record(reps, args.steps)
        legs["none"]["host_enqueue_s"] = host_enqueue[0]
        kernel_us_runs = [k * 1e3 / args.steps for _, k in reps]
    # the same launches replaying ONE action tensor (what rounds 1-2 timed): its reads may be cache hits
    if args.only_leg in (None, "replayed"):
        run(2, None, rotate=False)
        reps = timed_reps(args.steps, None, events=True, rotate=False)
        replay_us_runs = [k * 1e3 / args.steps for _, k in reps]
        if el_none is None:
            el_none, legs["none"] = leg_record(reps, args.steps)
            kernel_us_runs = replay_us_runs
    elapsed, collective = el_none, "none" + (f" ({no_coll_why})" if no_coll_why else "")
    diag = None
    if dist is not None and args.only_leg is None:
        # ---- leg "last_row" = `value`: the collective of the path (north_star): after every launch ONE
        # all-gather of the local CURRENT observation shard ([N_local, ...]: 512 KiB per rank for cfg2) gives
        # every rank the concatenated observation tensor of all world * N envs
        g_last = [ObsGatherer(o[0][-1], world, dist, always_collective=True) for o in outs]
        run(max(args.warmup, 1), g_last)
        reps = timed_reps(args.steps, g_last)
        el_last, legs["last_row"] = leg_record(reps, args.steps)
        legs["last_row"]["host_enqueue_s"] = host_enqueue[0]
        legs["last_row"]["bytes_per_rank_per_launch"] = g_last[0].local.numel() * g_last[0].local.element_size()
        elapsed = el_last
        collective = ("all_gather_into_tensor (%s, %d rank%s, async_op) of the current observation shard after every "
                      "launch, on the backend's high-priority stream beside the next launch"
                      % ("RCCL" if args.backend == "nccl" else args.backend, dist.get_world_size(), "" if world == 1 else "s"))
        # ---- what a scaling line needs to explain itself: per rank, the launch alone (HIP events on
        # the launch stream, from the `none` leg), the collective alone (G back-to-back gathers between events, nothing
        # else on the device) and what the collective adds per launch when it runs beside the rollouts; min / max over ranks
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        G = 20
        barrier()
        e0.record()
        for j in range(G):
            g_last[j % NB].start().wait()
        e1.record()
        e1.synchronize()
        gather_us = e0.elapsed_time(e1) * 1e3 / G
        barrier()
        mine = {"launch_us": statistics.median(kernel_us_runs), "gather_alone_us": gather_us,
                "launch_plus_gather_us": el_last * 1e6 / args.steps,
                "added_per_launch_us": (el_last - el_none) * 1e6 / args.steps}
        if world > 1:
            allv = [None] * world
            dist.all_gather_object(allv, mine)
        else:
            allv = [mine]
        diag = {k: {"min": min(v[k] for v in allv), "max": max(v[k] for v in allv)} for k in mine}
        diag["per_rank"] = allv
        diag["note"] = ("launch_us: HIP events on the launch stream, no collective; gather_alone_us: the all-gather with nothing "
                        "else running (events around %d of them); added_per_launch_us: (last_row - none) / launches -- what is "
                        "left of the gather after overlap (the rollout grid holds every CU: a gather can only run between launches)" % G)
        del g_last
        # ---- leg "full": every observation of the rollout, [K, N_local, ...] per rank per launch
        full_bytes = outs[0][0].numel() * outs[0][0].element_size()
        if args.full_gather_steps > 0 and full_bytes * world * NB < (64 << 30):
            g_full = [ObsGatherer(o[0], world, dist, always_collective=True) for o in outs]
            run(2, g_full)
            ks = min(args.steps, args.full_gather_steps)
            el_full, _ = timed(ks, g_full)
            legs["full"] = {"elapsed_s": el_full, "steps": ks, "env_steps_per_s": world * N * F * ks / el_full,
                            "bytes_per_rank_per_launch": full_bytes}
            del g_full
        # ---- leg "peer_copy" (opt-in: --peer-copy; reported beside `value`, never `value`; runs LAST, when every number of the
        # contract line is final -- ADVICE r4): the same exchange without a collective KERNEL -- every rank's buffer mapped into
        # the others with hipIpc handles, the shard copied device to device on a side stream after every launch, a bounded flag
        # wait at the end (include/mdpp.h mdpp_peer_*, dist.PeerGatherer).  Every phase is agreed between the ranks before the
        # next one starts (the leg has collectives inside: every rank takes a phase, or none does).
        if args.peer_copy:
            try:
                from mdp_playground_amd.dist import PeerGatherer

                class _PeerWork:                        # (the interface run() expects of a collective's Work handle)
                    def __init__(self, g, ticket):
                        self.g, self.ticket = g, ticket

                    def is_completed(self):
                        return False

                    def wait(self):                     # buffer reuse: this rank's copies have left
                        self.g.fence(self.ticket)

                    def finish(self):                   # end of the leg: every rank's shard has arrived
                        self.g.wait(self.ticket)

                class _PeerStart:
                    def __init__(self, g):
                        self.g = g

                    def start(self):
                        return _PeerWork(self.g, self.g.start())

                def all_ranks_ok(flag):
                    if world == 1:
                        return bool(flag)
                    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return bool(t.item())

                def phase(fn):                          # run one phase locally, then agree: (ok on every rank, local error)
                    err = None
                    try:
                        fn()
                    except Exception as e:
                        err = repr(e)
                    return all_ranks_ok(err is None), err

                peers, res = [], {}

                def build():                            # local half: buffers + handles (a partial list is closed below)
                    for o in outs:
                        peers.append(PeerGatherer(o[0][-1], world, rank, dist, slots=2, defer_exchange=True))

                def exchange():                         # collective half: exchange the handles, map the peers' buffers
                    if world > 1:
                        for g in peers:
                            g.exchange(dist)

                def check():                            # the gathered bytes ARE the shards: the peer gather against RCCL's
                    peers[0].wait(peers[0].start())
                    ref = ObsGatherer(outs[0][0][-1], world, dist, always_collective=True)
                    w = ref.start()
                    w.wait()
                    torch.cuda.synchronize(device)
                    if not torch.equal(peers[0].out.view(torch.uint8).reshape(-1), ref.out.view(torch.uint8).reshape(-1)):
                        raise RuntimeError("peer-copied rows differ from the RCCL gather of the same buffer")

                def warm():
                    run(max(args.warmup, 1), [_PeerStart(g) for g in peers])

                def measure():
                    reps = timed_reps(args.steps, [_PeerStart(g) for g in peers])
                    res["el"], res["rec"] = leg_record(reps, args.steps)

                why = None
                for fn in (build, exchange, check, warm, measure):
                    ok, err = phase(fn)
                    if not ok:
                        why = "%s: %s" % (fn.__name__, err or "failed on another rank")
                        break
                if why is None:
                    st = [g.status() for g in peers]
                    legs["peer_copy"] = res["rec"]
                    legs["peer_copy"].update({"host_enqueue_s": host_enqueue[0], "bytes_per_rank_per_launch": peers[0].nbytes,
                                              "timeouts": sum(1 for x in st if x[0]), "finegrained_buffers": bool(st[0][1]),
                                              "checked_against_rccl": True,
                                              "what": "hipIpc-mapped buffers, hipMemcpyAsync device to device on a side stream per launch, "
                                                      "flag wait at the end of the timed region"})
                    if diag is not None:
                        diag["peer_copy_added_per_launch_us"] = (res["el"] - el_none) * 1e6 / args.steps
                else:
                    legs["peer_copy"] = {"error": why}
                for g in peers:
                    g.close()
                del peers
            except Exception as e:                      # a reported extra, never fatal for the contract line
                legs["peer_copy"] = {"error": repr(e)}
    total_steps = world * N * F * args.steps
    value = total_steps / elapsed
    mark("main_legs")

    # ---- roofline of the dominant kernel (fused rollout), per launch: the median repeat
    launch_us = statistics.median(kernel_us_runs)
    per_launch_s = launch_us * 1e-6
    alg_bytes = wl["alg_bytes_fused"] * N * F
    achieved = alg_bytes / per_launch_s / 1e9
    kname = env.rollout_kernel_name(F)          # what the library's dispatch launches (mdpp_kernel_name)
    traffic, traffic_src = committed_traffic(args.workload, args.rng, N, F, kname)
    main_pmc = (pmc or {}).get(leg_name(args.workload, args.rng))
    main_valu = None
    if main_pmc is not None and kname.split("<")[0] in main_pmc["kernels"]:   # measured in this run, on the kernel that was timed
        traffic, traffic_src = main_pmc["bytes_per_launch"], main_pmc["note"]
        main_valu = main_pmc.get("valu_insts_per_launch")
    vr = valu_roofline(main_valu, launch_us, achieved / HBM_PEAK_GBS)
    roofline = {"bound": "hbm", "bound_measured": vr["bound"], "valu_frac": vr["valu_frac"], "valu": vr, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kname,
                "alg_bytes_per_env_step": wl["alg_bytes_fused"], "alg_bytes_per_launch": alg_bytes,
                "launch_us": launch_us, "env_steps_per_launch": N * F,
                "launch_us_runs": {"min": min(kernel_us_runs), "median": launch_us, "max": max(kernel_us_runs), "repeats": R},
                "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                "achievable": f"{HBM_ACHIEVABLE_GBS:.0f} GB/s: the float4 copy of MI355X_MICROARCH.md; `peak_measured` is this device's own",
                "action_tensors": NA, "action_bytes_rotated": NA * acts[0].numel() * acts[0].element_size(),
                "reads": f"`frac`: the launches cycle through {NA} distinct action tensors "
                         f"({NA * acts[0].numel() * acts[0].element_size() >> 20} MiB together, above the 256 MiB "
                         "Infinity Cache); `frac_replayed`: the same launches replaying one tensor"}
    if replay_us_runs is not None:
        rus = statistics.median(replay_us_runs)
        roofline["frac_replayed"] = alg_bytes / (rus * 1e-6) / 1e9 / HBM_PEAK_GBS
        roofline["launch_us_replayed"] = rus
        roofline["launch_us_replayed_runs"] = {"min": min(replay_us_runs), "median": rus, "max": max(replay_us_runs)}
    if args.only_leg is not None:
        roofline["only_leg"] = args.only_leg

    single = None
    if not args.no_single_step:
        single = single_step_leg(env, wl, acts[0], N, device)
    env.close()
    del outs, acts
    mark("single_step")

    # ---- the other BASELINE configs under the same clock (one GPU; not part of `value`)
    workloads = None
    if extra and rank == 0:
        workloads = {}
        for w, r in extra:
            key = leg_name(w, r)
            try:
                leg = workload_leg(w, r, device, args.fuse, max(args.workload_steps, 10), 3)
                rec = (pmc or {}).get(key)
                if rec is not None and leg["kernel"].split("<")[0] in rec["kernels"]:
                    leg["traffic"], leg["traffic_source"] = rec["bytes_per_launch"], rec["note"]
                    leg.update(valu_roofline(rec.get("valu_insts_per_launch"), leg["launch_us"], leg["frac"]))
                else:
                    leg["traffic"], leg["traffic_source"] = committed_traffic(w, r, leg["envs"], leg["fuse"], leg["kernel"])
                workloads[key] = leg
            except Exception as e:          # reported extras, never fatal for the contract line
                workloads[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
    mark("workloads")
    if rank == 0:
        peaks = hbm_ceilings(device)
        roofline["peak_measured"] = peaks
        # the fraction that belongs to `value` (the leg WITH the path's all-gather; `frac` is the kernel alone)
        roofline["frac_value"] = wl["alg_bytes_fused"] * (value / world) / 1e9 / HBM_PEAK_GBS
        roofline["frac_of_measured_copy"] = achieved / peaks["copy_GBps"]
        roofline["frac_of_measured_write"] = achieved / peaks["write_GBps"]
        roofline["frac_of_measured_read"] = achieved / peaks["read_GBps"] if peaks.get("read_GBps") else None

    # who ran where (the line explains a multi-GPU run by itself): rank -> device of every rank, RCCL's version
    placement = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": torch.cuda.get_device_name(device),
                 "pid": os.getpid()}
    try:
        placement["device_uuid"] = str(torch.cuda.get_device_properties(device).uuid)
    except Exception:
        placement["device_uuid"] = None
    if dist is not None and world > 1:
        placements = [None] * world
        dist.all_gather_object(placements, placement)
    else:
        placements = [placement]
    try:
        rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        rccl_version = None
    if dist is not None and world > 1:      # every number that needs the group is final: leave it BEFORE the CPU baselines
        barrier()
        dist.destroy_process_group()
        dist_done = True
    else:
        dist_done = False
    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baselines-only", args.workload] +
                               (["--cpu-all-cores"] if args.cpu_all_cores else []),
                               capture_output=True, text=True, timeout=600)
            got = [ln for ln in r.stdout.splitlines() if ln.startswith("CPU_BASELINES ")]
            rec = json.loads(got[-1][len("CPU_BASELINES "):])
            cpu_py, cpu_py_all, cpu_all = rec["cpu_py"], rec["cpu_py_all"], rec["cpu_all"]
        except Exception as e:          # a reported extra, never fatal
            cpu_py_all = {"error": repr(e)}
    cpu_port = None
    if rank == 0 and not args.no_cpu_baseline and not (
            wl["kind"] == "continuous" and wl["config"].get("image_representations")):
        cpu_port = cpu_baseline(wl)      # (the C port has no timed picture path for continuous envs)
    mark("cpu_port_and_ceilings")

    if rank == 0:
        v_none = legs["none"]["env_steps_per_s"]
        v_last = legs["last_row"]["env_steps_per_s"] if "last_row" in legs else None
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if wl["kind"] == "discrete" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: BASELINE.json configs "
                                   f"({json.dumps(wl['config'], sort_keys=True)}), "
                                   f"{N} env instances per GPU, random actions, same-step autoreset, rng={args.rng}; "
                                   f"ONE bench step = one fused launch of {F} env steps of every instance "
                                   f"(= {world * N * F} env steps)",
                       "envs_per_gpu": N, "fuse": F, "env_steps_per_bench_step": world * N * F, "rng": args.rng,
                       "disabled_kernels": args.disable or None,
                       "collective": collective,
                       "value_is": "the `last_row` leg (launch + the path's all-gather) for every --gpus, one rank included"
                                   if v_last is not None else "the `none` leg (no process group in this run)"},
            "value_none": v_none, "value_last_row": v_last,
            "value_peer_copy": legs.get("peer_copy", {}).get("env_steps_per_s"),
            "value_runs": legs["last_row" if v_last is not None else "none"]["env_steps_per_s_runs"], "repeats": R,
            "value_is_median_of_runs": True,
            # False: `value` is NOT the leg with the path's collective (no process group could be made) -- do not
            # compare it with lines where it is
            "collective_ok": v_last is not None,
            "multi_rank_diagnostics": diag,
            "device_count": torch.cuda.device_count(), "rccl_version": rccl_version, "backend": backend_used,
            "rank_devices": placements,
            "roofline": roofline,
            "cpu_baseline": cpu_py if cpu_py is not None else cpu_port,
            "cpu_baseline_all_cores": cpu_py_all, "cpu_baseline_port": cpu_port,
            "cpu_baseline_port_all_cores": cpu_all,
            "single_step": single, "collective_legs": legs, "workloads": workloads,
            "launches": args.steps, "elapsed_s": elapsed,
        }
        line["config"]["workload_short"] = (f"{args.workload}: {SHORT.get(args.workload, wl['kind'])}; {N} envs/GPU, random actions, "
                                            f"same-step autoreset; 1 bench step = 1 fused launch of {F} env steps")
        line["config"]["collective_short"] = ("none" if v_last is None else
                                              f"all_gather_into_tensor({backend_used}, {world} rank{'s' if world != 1 else ''}) after every launch")
        line["timing_s"] = phases
        emit(line, args.detail_out, args.print_detail)
    if dist is not None and not dist_done:
        dist.destroy_process_group()


LINE_MAX_BYTES = 4096       # the driver reads a bounded tail of stdout: the contract line must fit it whole (BENCH_r05: 26 KB, parsed = null)

SHORT = {   # one clause per workload for `config.workload` (the full config dicts are in the detail record)
    "cfg2": "BASELINE configs[1]: discrete 8x8, delay 4, sequence_length 3",
    "cfg2_noise": "configs[1] + transition_noise 0.1 + reward_noise 0.1",
    "cfg3": "BASELINE configs[2]: continuous move_to_a_point, 4 relevant + 8 irrelevant dims, order 1",
    "cfg4": "BASELINE configs[3]: discrete 8x8, 84x84 polygon pictures, shift + rotate",
    "cfg5": "BASELINE configs[4] shard: continuous move_to_a_point, order 2, p/r noise 0.05",
    "img100_all": "discrete 8x8, 100x100 pictures, shift+scale+rotate+flip (the reference's image sweep)",
    "img100_shift": "discrete 8x8, 100x100 pictures, shift",
}


def _r(x, nd=4):
    """Numbers of the contract line at 4 significant-ish digits (the detail record keeps every bit)."""
    if x is None or isinstance(x, (bool, int, str)):
        return x
    try:
        return float(f"{float(x):.{nd}g}") if abs(x) < 1e4 else float(f"{float(x):.6g}")
    except (TypeError, ValueError):
        return None


def compact_line(full):
    """The driver's contract line from the full record: every contract key, `roofline` and `cpu_baseline` of the dominant kernel,
    the single-step figures and ONE short list per extra workload -- [launch_us, frac, valu_frac, traffic / algorithmic bytes,
    step() launch_us or null].  Prose, per-repeat lists and nested per-leg records stay in the detail record
    (`detail`: the file it was written to).  Always <= LINE_MAX_BYTES (tests/test_bench_host.py)."""
    roof = full.get("roofline") or {}
    cpu = full.get("cpu_baseline") or {}
    cfg = full.get("config") or {}
    single = full.get("single_step") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 6)
    line["config"] = {"workload": cfg.get("workload_short") or str(cfg.get("workload"))[:160],
                      "envs_per_gpu": cfg.get("envs_per_gpu"), "fuse": cfg.get("fuse"),
                      "env_steps_per_bench_step": cfg.get("env_steps_per_bench_step"), "rng": cfg.get("rng"),
                      "collective": cfg.get("collective_short") or str(cfg.get("collective"))[:80]}
    line["roofline"] = {"bound": roof.get("bound"), "achieved": _r(roof.get("achieved"), 5), "peak": roof.get("peak"),
                        "unit": roof.get("unit"), "frac": _r(roof.get("frac")), "traffic": roof.get("traffic"),
                        "frac_value": _r(roof.get("frac_value")), "frac_replayed": _r(roof.get("frac_replayed")),
                        "valu_frac": _r(roof.get("valu_frac")), "bound_measured": roof.get("bound_measured"),
                        "kernel": roof.get("kernel"), "alg_bytes_per_env_step": roof.get("alg_bytes_per_env_step"),
                        "alg_bytes_per_launch": roof.get("alg_bytes_per_launch"), "launch_us": _r(roof.get("launch_us"), 5),
                        "traffic_source": "pmc, this run" if "in this run" in str(roof.get("traffic_source")) else roof.get("traffic_source")}
    if cpu:
        line["cpu_baseline"] = {"value": _r(cpu.get("value"), 6), "unit": cpu.get("unit"), "cores": cpu.get("cores"),
                                "kind": cpu.get("kind"), "form": cpu.get("form"), "sample": str(cpu.get("sample_short") or cpu.get("sample"))[:140],
                                "reference_equivalent": _r(((cpu.get("reference_equivalent") or {}).get("value")), 6)}
    else:
        line["cpu_baseline"] = None
    port = full.get("cpu_baseline_port") or {}
    line["cpu_baseline_port"] = {"value": _r(port.get("value"), 6), "cores": port.get("cores")} if port.get("value") else None
    line["value_none"], line["value_last_row"] = _r(full.get("value_none"), 6), _r(full.get("value_last_row"), 6)
    line["collective_ok"] = full.get("collective_ok")
    line["single_step"] = {"launch_us_events": _r(single.get("launch_us_events")),
                           "graph_us_per_step": _r((single.get("graph") or {}).get("us_per_step_events")),
                           "kernel": single.get("kernel")} if single else None
    wls = full.get("workloads")
    if wls:
        out = {}
        for k, v in wls.items():
            if "error" in v:
                out[k] = "error"
                continue
            ratio = (v["traffic"] / v["alg_bytes_per_launch"]) if v.get("traffic") and v.get("alg_bytes_per_launch") else None
            st = (v.get("single_step") or {}).get("launch_us_events")
            out[k] = [_r(v.get("launch_us")), _r(v.get("frac"), 3), _r(v.get("valu_frac"), 3), _r(ratio, 3), _r(st, 3)]
        line["workloads"] = out
        line["workloads_fields"] = "launch_us, hbm frac, valu_frac, pmc traffic / algorithmic, step() us"
    line["detail"] = full.get("detail_file")
    line["launches"], line["elapsed_s"] = full.get("launches"), _r(full.get("elapsed_s"), 6)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_MAX_BYTES:        # never the case for this file's own legs; a longer `workloads` map goes first
        for k in ("workloads", "workloads_fields", "single_step", "cpu_baseline_port"):
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= LINE_MAX_BYTES:
                break
    return s


def emit(full, detail_out=None, print_detail=False):
    """Write the full record to `detail_out`, flush C stdio (RCCL prints its version banner through printf into a
    buffer that would otherwise land AFTER everything Python printed, at exit), then print the contract line: the LAST line of stdout."""
    if detail_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_out)), exist_ok=True)
            with open(detail_out, "w") as f:
                json.dump(full, f)
            full["detail_file"] = os.path.relpath(detail_out, ROOT) if os.path.abspath(detail_out).startswith(ROOT) else detail_out
        except OSError:
            full["detail_file"] = None
    if print_detail:
        print("BENCH_DETAIL " + json.dumps(full), flush=True)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def _pmc_blocked():
    """Never from inside a profiled run: a child `rocprofv3 --pmc` would inherit the tracing environment of the
    `rocprofv3 --kernel-trace ... -- python bench.py` above it (counters + trace domains in one process)."""
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


def _run_group(cmd, timeout, **kw):
    """subprocess.run for a child that has children of its own (rocprofv3 -> python): its own session, and on a timeout
    the whole process group is ended, not only the direct child.  Returns the exit code, -999 on a timeout."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True, **kw)
    try:
        return p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return -999


def live_traffic_all(specs, launches=3, timeout=300):
    """HBM bytes per fused launch from PMC counters, measured NOW, for every (workload, rng, envs, fuse) of `specs`:
    two child rocprofv3 runs (`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: they do not fit one pass; no trace domain
    beside them) of tools/pmc_workloads.py, which launches the same fused rollouts workload after workload with a
    marker dispatch between them; read bytes = 2 x FETCH_SIZE KB (gfx950 tallies 128-B read requests at 64 B,
    MI355X_MICROARCH.md), write bytes = WRITE_SIZE KB, summed over every mdpp:: kernel of the rollout call (reset
    kernels excluded).  Returns {leg name: {"bytes_per_launch", "kernels", "note"}} or None when rocprofv3 is not
    usable here (the committed record is used instead)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or _pmc_blocked():
        return None
    tmp = tempfile.mkdtemp(prefix="mdpp_pmc_", dir="/tmp")
    seg = [{"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0, "kernels": {}} for _ in specs]
    valu_ok = True
    try:
        # (third pass: vector instructions issued -- the second roofline of the kernels that are issue-bound, SURVEY.md 8d
        #  "secondary ceilings"; a failure of THIS pass only drops `valu_frac`)
        for counter in PMC_PASSES:
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "tools", "pmc_workloads.py"), str(launches)] + \
                  [f"{w}:{r}:{n}:{f}" for w, r, n, f in specs]
            if _run_group(cmd, timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp")) != 0:
                if counter == "SQ_INSTS_VALU":
                    valu_ok = False
                    continue
                return None
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows += [row for row in csv.DictReader(open(f)) if row.get("Counter_Name") == counter]
            rows.sort(key=lambda row: int(row.get("Dispatch_Id") or 0))
            k = 0
            for row in rows:
                name = row["Kernel_Name"].split("(")[0]
                if "k_philox_normals" in name:          # the marker: next workload
                    k += 1
                    continue
                if k >= len(specs) or "mdpp::" not in name or "reset" in name:
                    continue
                seg[k][counter] += float(row["Counter_Value"])
                seg[k]["kernels"].setdefault(name, 0)
                if counter == "WRITE_SIZE":
                    seg[k]["kernels"][name] += 1
            if k != len(specs):
                if counter == "SQ_INSTS_VALU":
                    valu_ok = False
                    continue
                return None
        res = {}
        for (w, r, n, f), s in zip(specs, seg):
            if not s["kernels"]:
                continue
            res[leg_name(w, r)] = {
                "bytes_per_launch": int(round((2.0 * s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024.0 / launches)),
                "kernels": " ".join(s["kernels"]),
                "valu_insts_per_launch": (s["SQ_INSTS_VALU"] / launches) if (valu_ok and s["SQ_INSTS_VALU"] > 0) else None,
                "note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, x2 on FETCH_SIZE), every mdpp:: "
                        f"kernel of {launches} fused launches, in this run"}
        return res or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_traffic(workload, rng, N, F, launches=3, timeout=75):
    """One workload's record of live_traffic_all: (bytes per launch, kernels, note) or None."""
    res = live_traffic_all([(workload, rng, N, F)], launches, timeout)
    rec = (res or {}).get(leg_name(workload, rng))
    return None if rec is None else (rec["bytes_per_launch"], rec["kernels"], rec["note"])


# (one rocprofv3 pass per counter: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md); SQ_INSTS_VALU rides in a
#  third -- the three run beside the CPU baselines since round 6, so the third pass is no longer on the driver's clock)
PMC_PASSES = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")
VALU_CLOCK_HZ = 2.4e9       # MI355X peak engine clock (MI355X_MICROARCH.md); a wave64 vector instruction holds its SIMD for >= 4 cycles
N_SIMDS = 1024              # 256 CUs x 4 SIMDs


def valu_roofline(valu_insts_per_launch, launch_us, hbm_frac):
    """The second roofline of a launch (SURVEY.md 8d "secondary ceilings: integer ALU"): `valu_frac` = the share of the launch
    during which every SIMD's vector ALU is issuing, at the FLOOR of 4 cycles per wave instruction (float64, 32-bit integer
    multiplies and transcendentals take longer: the true share is higher) = SQ_INSTS_VALU / 1 024 SIMDs x 4 cycles / 2.4 GHz
    / launch time.  `bound` = which of the two fractions is the larger one: a kernel at 0.30 of HBM and 0.90 of its issue
    slots is an instruction-count problem, not a memory one."""
    if not valu_insts_per_launch or not launch_us:
        return {"valu_frac": None, "bound": "hbm"}
    valu_us = valu_insts_per_launch / N_SIMDS * 4.0 / VALU_CLOCK_HZ * 1e6
    vf = valu_us / launch_us
    return {"valu_frac": vf, "valu_insts_per_simd_per_launch": valu_insts_per_launch / N_SIMDS, "valu_issue_us_floor": valu_us,
            "bound": "valu" if vf > hbm_frac else "hbm"}


def committed_traffic(workload, rng, N, F, kname):
    """HBM bytes per launch from PMC counters.  They cannot be collected inside this process (rocprofv3
    wraps the program), so they are collected offline with tools/pmc_traffic.sh (FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 FETCH_SIZE x2 correction, MI355X_MICROARCH.md) and committed
    under profiles/; a record is used only for the launch shape AND kernel it was measured on."""
    tag = workload if rng == "numpy" else f"{workload}_{rng}"
    for name in (f"r03_traffic_{tag}.json", f"r02_traffic_{tag}.json", f"r02_traffic_{tag}_pipe.json", f"r01_traffic_{tag}.json"):
        tfile = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tfile):
            tfile = os.path.join(ROOT, "profiles", "archive", name)      # (rounds 1-3)
        if not os.path.exists(tfile):
            continue
        t = json.load(open(tfile))
        if t.get("envs") != N:
            continue
        kernels = " ".join(t.get("per_kernel_KB", {})) + " " + str(t.get("kernel", ""))
        if kname.split("<")[0] not in kernels:      # measured on another kernel (or the record does not say)
            continue
        if t.get("fuse") == F and "traffic_bytes_per_launch" in t:
            return t["traffic_bytes_per_launch"], os.path.basename(tfile)
        if "traffic_bytes_per_env_step" in t:
            # measured per env step over whole rollouts; the per-step traffic of a fused rollout does
            # not depend on the rollout length beyond the per-launch state I/O (a few bytes per env)
            return int(round(t["traffic_bytes_per_env_step"] * N * F)), os.path.basename(tfile)
    return None, None


def single_step_leg(env, wl, acts, N, device, n1=500, reps=20):
    """The one-launch-per-step API (mdpp_step) and a replayed HIP graph of such steps (`graph.exact`: how the replay keeps
    the handle's step counter -- 1 by value, 2 through the device-side tick offset: Philox streams, delay lines)."""
    a1 = acts[0].contiguous()
    for _ in range(50):
        env.step(a1)
    torch.cuda.synchronize(device)
    env.timer_begin()
    t1 = time.perf_counter()
    for _ in range(n1):
        env.step(a1)
    host1 = time.perf_counter() - t1         # the host's part: n1 calls enqueued (nothing waited for yet)
    ms1 = env.timer_end()
    torch.cuda.synchronize(device)
    wall1 = time.perf_counter() - t1
    # ... and a short burst into an EMPTY queue: over hundreds of calls the host is throttled by the queue once the device is
    # the slower side (then host_enqueue_us ~ launch_us_events although the device sets the pace); 48 calls are not
    burst = []
    for _ in range(5):
        torch.cuda.synchronize(device)
        tb = time.perf_counter()
        for _ in range(48):
            env.step(a1)
        burst.append((time.perf_counter() - tb) * 1e6 / 48)
    torch.cuda.synchronize(device)
    host_burst = min(burst)
    b1 = wl["alg_bytes_step"] * N
    single = {"env_steps_per_s": N * n1 / wall1, "launch_us_events": ms1 * 1e3 / n1,
              "host_enqueue_us": host1 * 1e6 / n1, "host_enqueue_us_burst": host_burst,
              "host_bound": bool(host_burst > 0.9 * ms1 * 1e3 / n1),
              "kernel": env.rollout_kernel_name(1),
              "alg_bytes_per_env_step": wl["alg_bytes_step"],
              "hbm_frac_events": b1 / (ms1 / 1e3 / n1) / 1e9 / HBM_PEAK_GBS}
    try:        # the device's own floor for one launch per step: empty kernels of the same grid, launched from C back to back
        import ctypes
        hu, du = ctypes.c_float(0.0), ctypes.c_float(0.0)
        rc = env._lib.mdpp_probe_launch(2000, max(1, (N + 63) // 64), ctypes.c_void_p(env._raw_stream()), ctypes.byref(hu), ctypes.byref(du))
        if rc == 0:
            single["launch_floor"] = {"what": "2000 launches of an EMPTY kernel with this grid, from C, back to back (mdpp_probe_launch)",
                                      "host_us_per_launch": hu.value, "device_us_per_launch": du.value}
            single["x_launch_floor"] = single["launch_us_events"] / max(hu.value, du.value)
    except Exception as e:        # a reported extra, never fatal
        single["launch_floor"] = {"error": repr(e)}
    if hasattr(env, "step_graph"):
        try:
            KG = 64
            g = env.step_graph(acts[:KG].contiguous())
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t2 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize(device)
            wall2 = time.perf_counter() - t2
            single["graph"] = {"steps_per_graph": KG, "exact": 2 if getattr(g, "_by_offset", False) else 1,
                               "env_steps_per_s": N * KG * reps / wall2,
                               "us_per_step_events": e0.elapsed_time(e1) * 1e3 / (KG * reps),
                               "hbm_frac_events": b1 / (e0.elapsed_time(e1) * 1e-3 / (KG * reps)) / 1e9 / HBM_PEAK_GBS}
        except Exception as e:        # a reported extra, never fatal
            single["graph"] = {"error": repr(e)}
    return single


def fold_reference_ratio(rec, workload):
    """The reference's OWN rate next to the restatement's: `reference_equivalent` = value x (reference : restatement speed ratio
    of this workload, both timed in one process on one core of the build container by tools/refgen/bench_reference.py ->
    profiles/py_baseline_ratio.json; the reference's files cannot travel to the GPU box).  None when the ratio file has no
    entry for the workload."""
    if not isinstance(rec, dict) or "value" not in rec:
        return rec
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "py_baseline_ratio.json")) as f:
            r = json.load(f)
        w = r["workloads"][workload]
        rec = dict(rec)
        rec["reference_equivalent"] = {"value": rec["value"] * w["ratio_reference_over_restatement"], "unit": rec["unit"],
                                       "ratio_reference_over_restatement": w["ratio_reference_over_restatement"],
                                       "ratio_measured_on": r.get("host"), "ratio_source": "profiles/py_baseline_ratio.json"}
    except Exception:
        rec = dict(rec)
        rec["reference_equivalent"] = None
    return rec


def cpu_baselines_forked(workload, all_cores=False):
    """Everything that forks worker processes (must run before this process initialises the GPU):
    the pure-Python restatement on 1 core and (all_cores) on all cores, and the C port on all cores."""
    wl = WORKLOADS[workload]
    cpu_py = cpu_py_all = cpu_all = None
    try:
        from baseline import bench_py
        cpu_py, cpu_py_all = bench_py.measure(wl["config"], wl["kind"], all_cores=all_cores)
        cpu_py, cpu_py_all = fold_reference_ratio(cpu_py, workload), fold_reference_ratio(cpu_py_all, workload)
    except Exception as e:          # a reported extra, never fatal
        cpu_py = None
        cpu_py_all = {"error": repr(e)}
    if (all_cores and wl["kind"] == "discrete" and not wl["config"].get("image_representations")
            and not wl["config"].get("irrelevant_features")):
        try:
            cpu_all = cpu_baseline_all_cores(workload)
        except Exception as e:
            cpu_all = {"error": repr(e)}
    return cpu_py, cpu_py_all, cpu_all


if __name__ == "__main__":
    main()
