"""Two ranks of one job, closed loop on env OUTPUT: each rank steps its shard
(dist.ShardedVectorEnv: global env ids [rank * N / 2, (rank + 1) * N / 2), observations assembled by ONE all-gather per
step / rollout) and holds the gathered observation tensor against a plain one-process RLToyVectorEnv of all N envs, bit
for bit -- BASELINE cfg2 and cfg5, numpy-exact and Philox streams, single steps and fused rollouts, rewards and flags of
the own shard too.  Started as `python -m torch.distributed.run --nproc-per-node 2 tests/_dist2_child.py` by
tests/test_gpu_dist.py with both ranks on the box's one GPU (backend gloo: RCCL refuses two ranks on one device; the
collective call is the same all_gather_into_tensor).  Rank 0 prints DIST2_OK.

Round 5: the same at WORLD SIZE 8 (`--nproc-per-node 8`, DIST_CHILD_N=65536, DIST_CHILD_CASES=cfg5): 8 x 8 192 envs against
one 65 536-env run -- BASELINE configs[4]'s sharding with the shard size the one GPU of the box can hold eight of: "a 1-GPU
run and an 8-GPU run produce identical trajectories" (SURVEY.md 8e)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mdp_playground_amd import RLToyVectorEnv  # noqa: E402
from mdp_playground_amd.dist import PeerGatherer, ShardedVectorEnv, shard_bounds  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world in (2, 8), world
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
           delay=4, sequence_length=3, seed=0)
ccfg = dict(state_space_type="continuous", state_space_dim=12, relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
            target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10, action_space_max=1,
            transition_dynamics_order=2, inertia=1, time_unit=0.1, transition_noise=0.05, reward_noise=0.05,
            make_denser=True, reward_function="move_to_a_point", seed=0)
N, T, K = int(os.environ.get("DIST_CHILD_N", "2048")), 12, 64
want = os.environ.get("DIST_CHILD_CASES", "cfg2,cfg5").split(",")
lo, hi = shard_bounds(N, rank, world)
assert (lo, hi) == (rank * (N // world), (rank + 1) * (N // world))
for name, c, rng in (("cfg2", cfg, "numpy"), ("cfg2", cfg, "philox"), ("cfg5", ccfg, "numpy"), ("cfg5", ccfg, "philox")):
    if name not in want:
        continue
    kw = dict(rng="philox", philox_seed=77) if rng == "philox" else {}
    sh = ShardedVectorEnv(N, rank, world, dist, device=dev, always_collective=True, autoreset="same_step", **kw, **c)
    whole = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", **kw, **c)
    g = torch.Generator(device=dev)
    g.manual_seed(5)                                           # the same global actions on both ranks
    o_s, _ = sh.reset()
    o_w, _ = whole.reset()
    assert o_s.shape == o_w.shape and torch.equal(o_s, o_w), (name, rng, "reset")
    for t in range(T):
        if c is cfg:
            a = torch.randint(0, 8, (N,), generator=g, device=dev, dtype=torch.int32)
        else:
            a = torch.rand((N, 12), generator=g, device=dev) * 2 - 1
        go, r1, te1, tr1, _ = sh.step(a[lo:hi].contiguous())
        o2, r2, te2, tr2, _ = whole.step(a)
        assert torch.equal(go, o2), (name, rng, t, "gathered observations")
        assert torch.equal(r1, r2[lo:hi]) and torch.equal(te1, te2[lo:hi]) and torch.equal(tr1, tr2[lo:hi]), (name, rng, t)
    if c is cfg:
        acts = torch.randint(0, 8, (K, N), generator=g, device=dev, dtype=torch.int32)
    else:
        acts = torch.rand((K, N, 12), generator=g, device=dev) * 2 - 1
    # the same gather as peer copies through the C-ABI (mdpp_peer_*: each rank's buffer mapped into the other PROCESS with
    # hipIpc handles, device-to-device copies on a side stream, a bounded flag wait) -- two processes, one device here
    pg = PeerGatherer(sh._last, world, rank, dist, slots=2)             # (rollout() leaves its last observation row there)
    for rep in range(3):                                       # (more pushes than slots: the buffers are reused)
        ob, rw, te, tr, glob = sh.rollout(acts[:, lo:hi].contiguous())
        ob2, rw2, te2, tr2 = whole.rollout(acts)
        assert torch.equal(ob, ob2[:, lo:hi]) and torch.equal(rw, rw2[:, lo:hi]) and torch.equal(te, te2[:, lo:hi]), (name, rng)
        assert torch.equal(glob, ob2[-1]), (name, rng, "gathered last row")
        peer = pg.wait(pg.start()).flatten(0, 1)
        torch.cuda.synchronize()
        assert pg.status()[0] == 0, (name, rng, "a rank's shard never arrived")
        assert torch.equal(peer, ob2[-1]), (name, rng, "peer-copy gather")
        dist.barrier()                                         # (nobody pushes into a slot a peer is still comparing)
    pg.close()
    assert int(sh.env.status().sum()) == 0
    sh.close(); whole.close()
    dist.barrier()
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("DIST2_OK", flush=True)
