"""Multi-GPU sharding of a batch of env instances: one process per GPU, no data-path
collective inside step() (env instances are independent), and ONE all-gather of the local
observation shard per step/rollout to give every rank the concatenated observation tensor
(RCCL over xGMI with backend "nccl"; gloo in the CPU tests).

Sharding is by contiguous global env id: rank r of R owns ids [r*N/R, (r+1)*N/R).  Env i's
streams are keyed by its GLOBAL id (RLToyVectorEnv(env_id_offset=...)), so a 1-GPU run and an
R-GPU run of the same job produce identical trajectories.
"""
from __future__ import annotations

import torch


def shard_bounds(total_envs: int, rank: int, world: int):
    """[lo, hi) of the global env ids owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ObsGatherer:
    """all_gather_into_tensor of a fixed local buffer into a preallocated [world, ...] buffer.

    The gathered tensor is laid out rank-major: out[r] is rank r's shard, so for single-step
    buffers of shape [N_local, ...] `out.flatten(0, 1)` is the [N_global, ...] observation
    tensor in global env-id order; for rollout buffers [K, N_local, ...] use
    `out.transpose(0, 1).flatten(1, 2)` for [K, N_global, ...]."""

    def __init__(self, local: torch.Tensor, world: int, dist_module=None, group=None, always_collective=False):
        self.local = local
        self.always_collective = bool(always_collective)   # one rank: still go through the backend (bench, tests)
        self.world = int(world)
        self.dist = dist_module
        self.group = group
        # concatenation form along dim 0 (accepted by both RCCL and gloo), viewed rank-major
        self._flat = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]),
                                 dtype=local.dtype, device=local.device)
        self.out = self._flat.view((self.world,) + tuple(local.shape))

    def __call__(self) -> torch.Tensor:
        if self.dist is None or (self.world == 1 and not self.always_collective):
            self.out[0].copy_(self.local)
            return self.out
        self.dist.all_gather_into_tensor(self._flat, self.local, group=self.group)
        return self.out

    def start(self):
        """The same gather, asynchronous with respect to the CURRENT stream: the backend's own stream waits for what
        the current stream has enqueued so far (the rollout that fills `local`) and runs the collective beside
        whatever the current stream does next; nothing is put back into the current stream.  Returns the Work handle
        (`.is_completed()`, `.wait()`; `self.out` holds the result once it is done), or None when no backend is
        involved (the copy then runs on the current stream)."""
        if self.dist is None or (self.world == 1 and not self.always_collective):
            self.out[0].copy_(self.local)
            return None
        return self.dist.all_gather_into_tensor(self._flat, self.local, group=self.group, async_op=True)


class ShardedVectorEnv:
    """Convenience wrapper: builds this rank's shard of a `total_envs` job and returns globally
    gathered observations from step()/rollout().  Rewards/flags stay local (each rank acts on
    its own shard); pass gather_all=True to gather them too."""

    def __init__(self, total_envs, rank, world, dist_module=None, device=None, always_collective=False, **kwargs):
        from .vector_env import RLToyVectorEnv
        self.rank, self.world, self.dist = rank, world, dist_module
        lo, hi = shard_bounds(total_envs, rank, world)
        if (hi - lo) * world != total_envs:
            raise ValueError("total_envs must divide evenly over the ranks (all_gather needs equal shards)")
        self.lo, self.hi = lo, hi
        self.env = RLToyVectorEnv(num_envs=hi - lo, device=device, env_id_offset=lo, **kwargs)
        self._always = bool(always_collective)
        self._g_obs = ObsGatherer(self.env._obs, world, dist_module, always_collective=self._always)
        # rollouts: ONE persistent staging row and gather output (a gatherer cached per rollout buffer pinned
        # up to nine whole [K, N, ...] buffers, and env.rollout(out=None) allocates a fresh one per call)
        self._last = torch.empty_like(self.env._obs)
        self._g_last = ObsGatherer(self._last, world, dist_module, always_collective=self._always)

    def reset(self, seed=None):
        obs, info = self.env.reset(seed=seed)
        return self._g_obs().flatten(0, 1), info

    def step(self, local_actions):
        obs, rew, term, trunc, info = self.env.step(local_actions)
        return self._g_obs().flatten(0, 1), rew, term, trunc, info

    def rollout(self, local_actions, out=None):
        """K fused steps of this rank's shard in one launch, then ONE all-gather of the rollout's LAST observation
        row: (obs_local [K, N_local, ...], reward, terminated, truncated, obs_global [N_global, ...]).  Per-step
        observations stay on their rank (gathering all of them is bound by the xGMI links).
        `obs_global` is a view of this object's one gather buffer: the next rollout() overwrites it."""
        obs, rew, term, trunc = self.env.rollout(local_actions, out)
        self._last.copy_(obs[-1])
        return obs, rew, term, trunc, self._g_last().flatten(0, 1)

    def close(self):
        self.env.close()
