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


class _DevBuf:
    """A device buffer the library owns, seen through __cuda_array_interface__ (torch.as_tensor makes a view, no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PeerGatherer:
    """The same gather as ObsGatherer -- every rank gets all ranks' shards of `local`, rank-major -- as device-to-device
    copies on the copy engines instead of a collective kernel (include/mdpp.h "mdpp_peer_*"): the rollout kernels hold
    every compute unit, so RCCL's all-gather kernel runs BETWEEN two launches; a peer copy runs beside them.

    One process per rank; the buffers' hipIpc handles are exchanged once through `dist_module.all_gather_object`.
    start() pushes the current contents of `local` (behind what the current stream has enqueued) into every rank's
    buffer; wait() makes the current stream wait until every rank's shard of that push has landed here; `out` is the
    [world, *local.shape] tensor (a view of the library's buffer).  Pushes alternate over `slots` buffers.  A rank that
    never arrives shows up as a status bit (status()), not as a hang.

    THE CONTRACT (nothing flows back from a reader to the writers -- ADVICE r4): every rank must wait() for push s, and
    have enqueued whatever reads `out` on the same stream, BEFORE it start()s push s + slots; a rank that runs `slots` or
    more pushes ahead of a peer overwrites rows the peer has not read, and the monotone flags would not show it.  close()
    frees / un-maps the buffers: callers synchronise the ranks (a barrier) before it, and views of `out` die with it.
    Not validated across a device boundary on the build pool (one GPU per box): experimental beside ObsGatherer, which is
    the path's collective (RCCL, what north_star names)."""

    _ITEM = {1: "|u1", 2: "<u2", 4: "<u4", 8: "<u8"}

    def __init__(self, local: torch.Tensor, world: int, rank: int, dist_module=None, slots: int = 2, defer_exchange: bool = False):
        import ctypes as C
        from . import _capi as capi
        if not local.is_contiguous():
            raise ValueError("PeerGatherer: the local shard must be contiguous")
        self._capi, self._lib, self._C = capi, capi.load(), C
        self.local, self.world, self.rank, self.slots = local, int(world), int(rank), int(slots)
        self.nbytes = local.numel() * local.element_size()
        self._h = C.c_void_p()
        dev = local.device.index or 0
        rc = self._lib.mdpp_peer_create(dev, self.world, self.rank, self.nbytes, self.slots, C.byref(self._h))
        if rc:
            raise capi.MdppError(f"mdpp_peer_create failed ({rc})")
        mine = (C.c_uint8 * capi.PEER_HANDLE_BYTES)()
        self._check(self._lib.mdpp_peer_handle(self._h, mine), "mdpp_peer_handle")
        self._mine, self._opened = bytes(mine), self.world == 1
        if self.world > 1 and not defer_exchange:
            if dist_module is None:
                raise ValueError("PeerGatherer: world > 1 needs a process group to exchange the handles")
            self.exchange(dist_module)
        self._seq, self._slot = 0, 0
        typestr = self._ITEM[local.element_size()]
        self._views = []
        for s in range(self.slots):
            ptr = self._lib.mdpp_peer_buffer(self._h, s)
            raw = torch.as_tensor(_DevBuf(ptr, (self.world * local.numel(),), typestr), device=local.device)
            self._views.append(raw.view(local.dtype).view((self.world,) + tuple(local.shape)))
        self.out = self._views[0]

    def exchange(self, dist_module):
        """The COLLECTIVE half of the set-up (every rank must call it): all_gather_object of the buffers' handles, then
        each rank maps the others' buffers.  `defer_exchange=True` in the constructor leaves it to the caller, who can
        first agree across ranks that every local half succeeded (bench.py: a rank that failed alone must not leave the
        others inside a collective)."""
        C, capi = self._C, self._capi
        allh = [None] * self.world
        dist_module.all_gather_object(allh, self._mine)
        blob = (C.c_uint8 * (capi.PEER_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(allh))
        self._check(self._lib.mdpp_peer_open(self._h, blob), "mdpp_peer_open")
        self._opened = True

    def _check(self, rc, what):
        if rc:
            msg = self._lib.mdpp_peer_last_error(self._h)
            raise self._capi.MdppError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.local.device).cuda_stream)

    def start(self):
        """Push the shard (asynchronous with respect to the current stream); returns the ticket wait() takes."""
        self._seq += 1
        self._slot = self._seq % self.slots
        self._check(self._lib.mdpp_peer_push(self._h, self._slot, self.local.data_ptr(), self._seq, self._stream()), "mdpp_peer_push")
        return (self._slot, self._seq)

    def fence(self, ticket=None):
        """The current stream waits (an event, no kernel) until this rank's copies of that push have left: `local` may be
        overwritten after it."""
        slot, _ = ticket if ticket is not None else (self._slot, self._seq)
        self._check(self._lib.mdpp_peer_fence(self._h, slot, self._stream()), "mdpp_peer_fence")

    def wait(self, ticket=None):
        slot, seq = ticket if ticket is not None else (self._slot, self._seq)
        self._check(self._lib.mdpp_peer_wait(self._h, slot, seq, self._stream()), "mdpp_peer_wait")
        self.out = self._views[slot]
        return self.out

    def __call__(self):
        return self.wait(self.start())

    def status(self):
        st, fg = self._C.c_uint32(0), self._C.c_int(0)
        self._check(self._lib.mdpp_peer_status(self._h, self._C.byref(st), self._C.byref(fg)), "mdpp_peer_status")
        return int(st.value), bool(fg.value)

    def close(self):
        if self._h:
            self._views, self.out = [], None
            self._lib.mdpp_peer_destroy(self._h)
            self._h = None
