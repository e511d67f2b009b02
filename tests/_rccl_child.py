"""One RCCL rank (world size 1) in a fresh process: dist.ShardedVectorEnv -- this rank's shard of the job, observations
assembled by all_gather_into_tensor over backend "nccl" (= RCCL) -- against a plain RLToyVectorEnv of the same envs, and
a shard of a larger job against the same global env ids of a plain env (streams are keyed by the GLOBAL id).
Prints RCCL_OK.  Started by tests/test_gpu_dist.py through tests/_spawn_helper.py."""
import os
import socket
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mdp_playground_amd import RLToyVectorEnv  # noqa: E402
from mdp_playground_amd.dist import ObsGatherer, ShardedVectorEnv  # noqa: E402

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=8, action_space_size=8,
           delay=4, sequence_length=3, seed=0)
ccfg = dict(state_space_type="continuous", state_space_dim=12, relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
            target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10, action_space_max=1,
            transition_dynamics_order=2, inertia=1, time_unit=0.1, transition_noise=0.05, reward_noise=0.05,
            make_denser=True, reward_function="move_to_a_point", seed=0)
for name, c, rng in (("cfg2", cfg, "numpy"), ("cfg2", cfg, "philox"), ("cfg5", ccfg, "numpy"), ("cfg5", ccfg, "philox")):
    N, T = 4096, 24
    sh = ShardedVectorEnv(N, 0, 1, dist, device=dev, always_collective=True, autoreset="same_step", rng=rng, **c)
    pl = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", rng=rng, **c)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    assert torch.equal(sh.env._obs, pl._obs), name           # (the constructors' reset())
    o_s, _ = sh.reset()
    o_p, _ = pl.reset()
    assert o_s.shape[0] == N and torch.equal(o_s, o_p), name
    for t in range(T):
        if c is cfg:
            a = torch.randint(0, 8, (N,), generator=g, device=dev, dtype=torch.int32)
        else:
            a = torch.rand((N, 12), generator=g, device=dev) * 2 - 1
        go, r1, te1, tr1, _ = sh.step(a)
        o2, r2, te2, tr2, _ = pl.step(a)
        assert go.data_ptr() != sh.env._obs.data_ptr()          # the gathered tensor, not the local buffer
        assert torch.equal(go, o2) and torch.equal(r1, r2) and torch.equal(te1, te2), (name, rng, t)
    # a fused rollout, then the gather of its last row (what bench.py's `last_row` leg does)
    if c is cfg:
        acts = torch.randint(0, 8, (64, N), generator=g, device=dev, dtype=torch.int32)
    else:
        acts = torch.rand((64, N, 12), generator=g, device=dev) * 2 - 1
    ob, rw, te, tr, glob = sh.rollout(acts)
    ob2, rw2, te2, tr2 = pl.rollout(acts)
    assert torch.equal(ob, ob2) and torch.equal(rw, rw2) and torch.equal(te, te2) and torch.equal(glob, ob2[-1]), (name, rng)
    sh.close(); pl.close()

# the second half of a two-rank job (global ids 2048..4095) == those envs of the one-rank job
N = 4096
whole = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", **cfg)
half = RLToyVectorEnv(num_envs=N // 2, device=dev, autoreset="same_step", env_id_offset=N // 2, **cfg)
acts = torch.randint(0, 8, (40, N), device=dev, dtype=torch.int32)
ow = whole.rollout(acts)[0]
oh = half.rollout(acts[:, N // 2:].contiguous())[0]
assert torch.equal(ow[:, N // 2:], oh)
# the timing reduction of bench.py and a gather through the backend
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 1.5
loc = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
gth = ObsGatherer(loc, 1, dist, always_collective=True)
assert torch.equal(gth().flatten(0, 1), loc)
# the asynchronous form bench.py uses: the backend's stream runs the gather beside the caller's next launch
loc.mul_(2.0)
w = gth.start()
assert w is not None
w.wait()
torch.cuda.synchronize()
assert w.is_completed() and torch.equal(gth.out.flatten(0, 1), loc)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", flush=True)
