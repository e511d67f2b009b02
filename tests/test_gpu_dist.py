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


def _contract_line_and_detail(stdout, detail_path):
    """What the driver reads: ONE JSON line, the LAST line of stdout, <= 4 KiB, with every contract key; the legs, repeats and
    per-rank records behind it are in the detail file (--detail-out).  Returns the detail record after checking that the
    line's figures are the record's."""
    import json
    out_lines = [ln for ln in stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out_lines if ln.startswith('{"metric"')]
    assert len(lines) == 1 and out_lines[-1] == lines[0] and len(lines[0].encode()) <= 4096, stdout[-2000:]
    line = json.loads(lines[0])
    d = json.loads(open(detail_path).read())
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == d[k], k
    assert abs(line["value"] / d["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / d["ms_per_step"] - 1) < 1e-4
    assert line["config"]["envs_per_gpu"] == d["config"]["envs_per_gpu"]
    assert line["config"]["env_steps_per_bench_step"] == d["config"]["env_steps_per_bench_step"]
    assert line["roofline"]["bound"] == "hbm" and abs(line["roofline"]["frac"] / d["roofline"]["frac"] - 1) < 1e-3
    assert line["roofline"]["kernel"] == d["roofline"]["kernel"] and line["cpu_baseline"]["value"] > 0
    assert line["collective_ok"] == d["collective_ok"]
    return d


@pytest.mark.timeout(900)
def test_bench_multi_rank_path_with_two_ranks_on_one_gpu(spawn_fresh, tmp_path):
    """The N > 1 code path of bench.py end to end on hardware: `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2` with both ranks on the one GPU of the box (backend gloo -- RCCL refuses two ranks on one device; the one-rank
    RCCL path is the test above).  Checks what the driver reads: ONE JSON line from rank 0, n_gpus 2, value = the leg with
    the all-gather, weak scaling (65 536... here 4 096 envs per rank), the three legs and the max-over-ranks timing."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(HERE)
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
            "--envs", "4096", "--fuse", "64", "--backend", "gloo", "--no-single-step", "--full-gather-steps", "2", "--peer-copy",
            "--detail-out", str(tmp_path / "detail.json")]
    r = spawn_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=800)
    assert r["returncode"] == 0, r["stdout"][-2000:] + "\n" + r["stderr"][-6000:]
    d = _contract_line_and_detail(r["stdout"], tmp_path / "detail.json")
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["envs_per_gpu"] == 4096 and d["config"]["env_steps_per_bench_step"] == 2 * 4096 * 64
    assert set(d["collective_legs"]) == {"none", "last_row", "peer_copy", "full"}
    pc = d["collective_legs"]["peer_copy"]
    assert "error" not in pc and pc["timeouts"] == 0 and pc["env_steps_per_s"] > 0 and d["value_peer_copy"] == pc["env_steps_per_s"]
    assert pc["checked_against_rccl"] is True           # (the peer-copied rows equal the backend's gather of the same buffer)
    assert d["value"] == d["value_last_row"] == d["collective_legs"]["last_row"]["env_steps_per_s"] > 0
    assert d["value_none"] == d["collective_legs"]["none"]["env_steps_per_s"] >= 0.5 * d["value"]
    assert abs(d["value"] - 2 * 4096 * 64 * 4 / d["elapsed_s"]) <= 1e-6 * d["value"]
    assert d["workloads"] is None
    # round 4: rank 0 times the CPU baselines at N > 1 too; the line explains itself -- per-rank launch / gather timings,
    # the repeats behind the median, and whether `value` is the leg with the collective
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0
    assert d["collective_ok"] is True and d["repeats"] == 5 and len(d["value_runs"]) == 5
    assert min(d["value_runs"]) <= d["value"] <= max(d["value_runs"])
    diag = d["multi_rank_diagnostics"]
    assert len(diag["per_rank"]) == 2 and all(k in diag for k in ("launch_us", "gather_alone_us", "added_per_launch_us"))
    assert 0 < diag["launch_us"]["min"] <= diag["launch_us"]["max"] and diag["gather_alone_us"]["min"] > 0
    assert d["roofline"]["launch_us_runs"]["repeats"] == 5


@pytest.mark.timeout(900)
def test_two_ranks_gathered_observations_equal_one_rank_run(spawn_fresh):
    """the loop closed at world size 2 on env OUTPUT -- two ranks (gloo, both on the box's one GPU)
    step dist.ShardedVectorEnv and each asserts that the all-gathered observations equal a plain RLToyVectorEnv of all
    the job's envs bit for bit: cfg2 and cfg5, both RNGs, single steps and fused rollouts (tests/_dist2_child.py)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(HERE, "_dist2_child.py")]
    r = spawn_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=800)
    assert r["returncode"] == 0, r["stdout"][-3000:] + "\n" + r["stderr"][-6000:]
    assert "DIST2_OK" in r["stdout"]


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.timeout(1500)
def test_bench_world_8_rehearsal_on_one_gpu(spawn_fresh, tmp_path):
    """What the driver's SCALE run executes at N = 8, rehearsed with eight ranks on the box's one GPU (backend gloo: RCCL
    refuses several ranks on one device; everything else -- torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE, shards
    keyed by the global env id, barriers, max-over-ranks timing, ONE all-gather per launch, eight hipIpc handles in the
    peer-copy leg, CPU baselines after the process group is gone -- is the N = 8 path).  ONE JSON line, n_gpus 8, per-rank
    diagnostics of all eight ranks, the rank -> device map, no timeouts."""
    root = os.path.dirname(HERE)
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2",
            "--envs", "8192", "--fuse", "64", "--backend", "gloo", "--no-single-step", "--full-gather-steps", "2", "--peer-copy",
            "--detail-out", str(tmp_path / "detail.json")]
    r = spawn_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=1400)
    assert r["returncode"] == 0, r["stdout"][-2000:] + "\n" + r["stderr"][-6000:]
    d = _contract_line_and_detail(r["stdout"], tmp_path / "detail.json")
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == 8192
    assert d["config"]["env_steps_per_bench_step"] == 8 * 8192 * 64
    assert abs(d["value"] - 8 * 8192 * 64 * 4 / d["elapsed_s"]) <= 1e-6 * d["value"]
    assert d["collective_ok"] is True and d["value"] == d["value_last_row"] > 0
    diag = d["multi_rank_diagnostics"]
    assert len(diag["per_rank"]) == 8 and all(x["launch_us"] > 0 for x in diag["per_rank"])
    assert [x["rank"] for x in d["rank_devices"]] == list(range(8)) and d["backend"] == "gloo" and d["device_count"] >= 1
    assert len({x["pid"] for x in d["rank_devices"]}) == 8
    pc = d["collective_legs"]["peer_copy"]
    assert "error" not in pc and pc["timeouts"] == 0 and pc["checked_against_rccl"] is True
    assert d["cpu_baseline"] is not None and d["cpu_baseline"]["value"] > 0     # (measured by a child of rank 0 after the group was left)


@pytest.mark.timeout(1500)
def test_eight_ranks_of_8192_envs_equal_one_run_of_65536(spawn_fresh):
    """"a 1-GPU run and an 8-GPU run produce identical trajectories".  Eight ranks (gloo, one GPU) x 8 192 envs of
    BASELINE cfg5 (numpy-exact and Philox streams) against ONE process stepping all 65 536: gathered observations, own-shard
    rewards and flags, single steps and fused rollouts, bit for bit (tests/_dist2_child.py)."""
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(HERE, "_dist2_child.py")]
    r = spawn_fresh(argv, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "DIST_CHILD_N": "65536", "DIST_CHILD_CASES": "cfg5"}, timeout=1400)
    assert r["returncode"] == 0, r["stdout"][-3000:] + "\n" + r["stderr"][-6000:]
    assert "DIST2_OK" in r["stdout"]


@pytest.mark.timeout(300)
def test_bench_preflight_refuses_rccl_with_fewer_devices_than_ranks(spawn_fresh):
    """`--gpus 8 --backend nccl` on a box with one device: a one-line reason and a non-zero exit code, not a hang in the
    rendezvous (RCCL cannot put two ranks on one device)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-device node: the preflight has nothing to refuse")
    root = os.path.dirname(HERE)
    r = spawn_fresh([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], env={}, timeout=250)
    assert r["returncode"] != 0
    assert "needs 8 visible devices" in (r["stdout"] + r["stderr"])
