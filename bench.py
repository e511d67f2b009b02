#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched RLToyEnv.step() hot path on MI355X.

    python bench.py --gpus 1 --steps 8192 --warmup 1024
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): discrete 8 states x 8 actions, reward_delay 4,
sequence_length 3, 65 536 env instances per GPU sharing one MDP, uniform random actions
(synthetic, pre-generated on the device), same-step autoreset, numpy-exact PCG64 streams.
A "step" is one env step of every instance of every rank.  Steps run as fused rollouts of
--fuse steps per launch (mdpp_step_n); with N > 1 each launch is followed by ONE RCCL
all-gather that assembles the current global observation tensor on every rank (env ids are
sharded contiguously, weak scaling; the gather overlaps the next launch).  The single-launch-per-step path (mdpp_step) is reported beside it.

Rank 0 prints ONE JSON line: the driver contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

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
    # BASELINE.json configs[4] (per-GPU shard of the 524 288-env job)
    "cfg5": dict(kind="continuous", envs=65536, alg_bytes_fused=102, alg_bytes_step=350,
                 config=dict(state_space_type="continuous", state_space_dim=12,
                             relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
                             target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10,
                             action_space_max=1, transition_dynamics_order=2, inertia=1,
                             time_unit=0.1, transition_noise=0.05, reward_noise=0.05,
                             make_denser=True, reward_function="move_to_a_point", seed=0)),
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


def cpu_baseline(wl, seconds=12.0):
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


def cpu_baseline_all_cores(wl_name, seconds=4.0):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8192)
    ap.add_argument("--warmup", type=int, default=1024)
    ap.add_argument("--fuse", type=int, default=None,
                    help="env steps per fused launch (mdpp_step_n); default 512, and 2048 with more than one rank: "
                         "every launch is followed by the path's one all-gather, whose host-side enqueue "
                         "(events, stream switch, RCCL call) is of the order of a 512-step launch")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="env instances per GPU")
    ap.add_argument("--rng", default="numpy", choices=["numpy", "philox"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # all-core CPU line first: it forks one worker per core, which must happen before this
    # process has initialised the GPU runtime
    cpu_all = None
    wl0 = WORKLOADS[args.workload]
    if (rank == 0 and world == 1 and not args.no_cpu_baseline and wl0["kind"] == "discrete"
            and not wl0["config"].get("image_representations")
            and not wl0["config"].get("irrelevant_features")):
        try:
            cpu_all = cpu_baseline_all_cores(args.workload)
        except Exception as e:          # a reported extra, never fatal
            cpu_all = {"error": repr(e)}
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: RCCL, even for N = 1
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)

    from mdp_playground_amd import RLToyVectorEnv
    from mdp_playground_amd.dist import ObsGatherer

    wl = WORKLOADS[args.workload]
    N = args.envs or wl["envs"]
    fuse = args.fuse if args.fuse is not None else (2048 if world > 1 else 512)
    F = max(1, min(fuse, args.steps, wl.get("fuse_max", fuse)))
    env = RLToyVectorEnv(num_envs=N, device=device, env_id_offset=rank * N, rng=args.rng,
                         autoreset="same_step", **wl["config"])
    acts = make_actions(wl, F, N, device, 12345 + rank)
    outs = [env.alloc_rollout(F), env.alloc_rollout(F)]     # alternate, so a gather can trail a launch
    out = outs[0]
    # The collective of the path: after every rollout launch ONE all-gather of the
    # local CURRENT observation shard ([N_local, ...], 512 KiB per rank for cfg2) gives every rank
    # the concatenated observation tensor of all world*N envs.  It runs on its own stream and
    # overlaps the next launch; the per-step observations of a fused rollout stay on their rank.
    gathers, comm, ev_done = None, None, [None, None]
    if dist is not None:
        gathers = [ObsGatherer(o[0][-1], world, dist) for o in outs]
        comm = torch.cuda.Stream(device=device)

    def run(steps):
        left, launches = steps, 0
        cur = torch.cuda.current_stream(device)
        while left > 0:
            k = min(F, left)
            j = launches & 1
            if gathers is not None and ev_done[j] is not None:
                cur.wait_event(ev_done[j])          # the gather that read this buffer two launches ago
            if k == F:
                env.rollout(acts, outs[j])
            else:
                env.rollout(acts[:k], tuple(t[:k] for t in outs[j]))
            if gathers is not None:
                ev = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    if k == F:
                        gathers[j]()
                    else:
                        gathers[j].local = outs[j][0][k - 1]
                        gathers[j]()
                        gathers[j].local = outs[j][0][-1]
                    ev_done[j] = torch.cuda.Event()
                    ev_done[j].record(comm)
            left -= k
            launches += 1
        if comm is not None:
            cur.wait_stream(comm)
        return launches

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(max(args.warmup, 1))
    barrier()
    if gathers is None:
        env.timer_begin()
    t0 = time.perf_counter()
    launches = run(args.steps)
    if gathers is None:
        kernel_ms = env.timer_end()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier()
    if gathers is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # roofline leg without the collective: HIP events around plain launches on this rank
        env.timer_begin()
        launches_r = 0
        for _ in range(8):
            env.rollout(acts, out)
            launches_r += 1
        kernel_ms = env.timer_end() * (args.steps / (launches_r * F))
    total_steps = world * N * args.steps
    value = total_steps / elapsed

    # ---- roofline of the dominant kernel (fused rollout), per launch
    per_launch_s = (kernel_ms / 1e3) / (args.steps / F)
    alg_bytes = wl["alg_bytes_fused"] * N * F
    achieved = alg_bytes / per_launch_s / 1e9
    kname = env.rollout_kernel_name(F)
    # HBM bytes per launch from PMC counters: collected offline with rocprofv3 --pmc (separate
    # passes, gfx950 FETCH_SIZE correction applied) and committed under profiles/; only valid for
    # the exact launch shape it was measured on.
    traffic = None
    tfile = os.path.join(ROOT, "profiles", f"r01_traffic_{args.workload}.json")
    if os.path.exists(tfile):
        t = json.load(open(tfile))
        if t.get("envs") == N and t.get("fuse") == F and "traffic_bytes_per_launch" in t:
            traffic = t["traffic_bytes_per_launch"]
        elif t.get("envs") == N and "traffic_bytes_per_env_step" in t:
            # measured per env step over whole rollouts (tools/pmc_traffic.sh); per-step traffic of a
            # fused rollout does not depend on the rollout length beyond the per-launch state I/O
            traffic = int(round(t["traffic_bytes_per_env_step"] * N * F))
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                "alg_bytes_per_env_step": wl["alg_bytes_fused"],
                "launch_us": per_launch_s * 1e6, "env_steps_per_launch": N * F}

    single = None
    if not args.no_single_step:
        a1 = acts[0].contiguous()
        for _ in range(50):
            env.step(a1)
        torch.cuda.synchronize(device)
        n1 = 500
        env.timer_begin()
        t1 = time.perf_counter()
        for _ in range(n1):
            env.step(a1)
        ms1 = env.timer_end()
        torch.cuda.synchronize(device)
        wall1 = time.perf_counter() - t1
        b1 = wl["alg_bytes_step"] * N
        single = {"env_steps_per_s": N * n1 / wall1, "launch_us_events": ms1 * 1e3 / n1,
                  "alg_bytes_per_env_step": wl["alg_bytes_step"],
                  "hbm_frac_events": b1 / (ms1 / 1e3 / n1) / 1e9 / HBM_PEAK_GBS}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not (
            wl["kind"] == "continuous" and wl["config"].get("image_representations")):
        cpu = cpu_baseline(wl)      # (the C port has no timed picture path for continuous envs)
    env.close()

    if rank == 0:
        line = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if wl["kind"] == "discrete" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: BASELINE.json configs "
                                   f"({json.dumps(wl['config'], sort_keys=True)}), "
                                   f"{N} env instances per GPU, random actions, same-step autoreset, "
                                   f"fused rollout of {F} steps per launch, rng={args.rng}",
                       "envs_per_gpu": N, "fuse": F,
                       "collective": ("all_gather of the current observation shard after every launch, "
                                      "overlapped on a side stream") if gathers is not None else "none"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "single_step": single,
            "launches": launches, "elapsed_s": elapsed,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
