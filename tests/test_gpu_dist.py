"""RCCL on the GPU box (nothing the driver ran ever initialised backend "nccl"): one rank in a FRESH
process started by a helper that never touched the GPU (tests/_spawn_helper.py; never a re-exec of this process)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(600)
def test_rccl_one_rank_sharded_env_equals_plain_env(spawn_fresh):
    r = spawn_fresh([sys.executable, os.path.join(HERE, "_rccl_child.py")],
                    env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"}, timeout=500)
    assert r["returncode"] == 0, r["stdout"][-3000:] + "\n" + r["stderr"][-6000:]
    assert "RCCL_OK" in r["stdout"]


@pytest.mark.timeout(900)
def test_bench_multi_rank_path_with_two_ranks_on_one_gpu(spawn_fresh):
    """The N > 1 code path of bench.py end to end on hardware: `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2` with both ranks on the one GPU of the box (backend gloo -- RCCL refuses two ranks on one device; the one-rank
    RCCL path is the test above).  Checks what the driver reads: ONE JSON line from rank 0, n_gpus 2, value = the leg with
    the all-gather, weak scaling (65 536... here 4 096 envs per rank), the three legs and the max-over-ranks timing."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(HERE)
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
            "--envs", "4096", "--fuse", "64", "--backend", "gloo", "--no-single-step", "--full-gather-steps", "2"]
    r = spawn_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=800)
    assert r["returncode"] == 0, r["stdout"][-2000:] + "\n" + r["stderr"][-6000:]
    lines = [ln for ln in r["stdout"].splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r["stdout"][-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["envs_per_gpu"] == 4096 and d["config"]["env_steps_per_bench_step"] == 2 * 4096 * 64
    assert set(d["collective_legs"]) == {"none", "last_row", "full"}
    assert d["value"] == d["value_last_row"] == d["collective_legs"]["last_row"]["env_steps_per_s"] > 0
    assert d["value_none"] == d["collective_legs"]["none"]["env_steps_per_s"] >= 0.5 * d["value"]
    assert abs(d["value"] - 2 * 4096 * 64 * 4 / d["elapsed_s"]) <= 1e-6 * d["value"]
    assert d["workloads"] is None and d["cpu_baseline"] is None
