"""N > 1 path on CPU: two processes, gloo backend, the same ObsGatherer / shard layout the GPU
bench uses with RCCL (backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mdp_playground_amd.dist import ObsGatherer, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(total, rank, world)
    # a stand-in "observation shard": obs of global env id g at step k is g * 1000 + k
    K, D = 3, 4
    local = torch.zeros((K, hi - lo, D), dtype=torch.float32)
    for k in range(K):
        local[k] = (torch.arange(lo, hi, dtype=torch.float32) * 1000 + k)[:, None]
    g = ObsGatherer(local, world, dist)
    out = g()
    glob = out.transpose(0, 1).flatten(1, 2)          # [K, N_global, D]
    ok = True
    for k in range(K):
        exp = (torch.arange(total, dtype=torch.float32) * 1000 + k)[:, None].expand(total, D)
        ok = ok and torch.equal(glob[k], exp)
    # single-step layout: [N_local, D] -> [N_global, D]
    g1 = ObsGatherer(local[0].contiguous(), world, dist)
    ok = ok and torch.equal(g1().flatten(0, 1), (torch.arange(total, dtype=torch.float32) * 1000)[:, None].expand(total, D))
    # the asynchronous form (ObsGatherer.start -> Work): what bench.py's collective leg uses
    w = g1.start()
    w.wait()
    ok = ok and torch.equal(g1.out.flatten(0, 1), (torch.arange(total, dtype=torch.float32) * 1000)[:, None].expand(total, D))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == float(world)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for total, world in [(65536 * 8, 8), (10, 3), (7, 7), (5, 8)]:
        spans = [shard_bounds(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


@pytest.mark.timeout(180)
def test_two_process_gloo_all_gather():
    world, total = 2, 64
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    assert ret[0] and ret[1]
