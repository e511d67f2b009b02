"""Host-side pieces of bench.py that run without a GPU: the committed PMC traffic record is picked for
the launch shape AND kernel it was measured on, and a plain `python bench.py --gpus N` starts its ranks
as a CHILD torch.distributed.run (never re-executing a process that has touched the GPU)."""
import sys
import types

import bench


def test_committed_traffic_is_keyed_by_shape_and_kernel():
    t, src = bench.committed_traffic("cfg2", "numpy", 65536, 512, "k_discrete_rollout_lean<OBS64=1,DELAY=1,HASMAX=0,EVN=1>")
    assert src == "r03_traffic_cfg2.json" and 6.0e8 < t < 6.4e8                 # 18.6 B x 65 536 x 512 (the newest record)
    t, src = bench.committed_traffic("cfg2", "numpy", 65536, 512, "k_discrete_rollout_pipe<OBS64=1,POW2=1,DELAY=1,S8=1>")
    assert src == "r02_traffic_cfg2_pipe.json" and 6.0e8 < t < 6.3e8            # the record of that kernel
    assert bench.committed_traffic("cfg2", "numpy", 65536, 512, "k_discrete_step<PHILOX=0>")[0] is None     # another kernel
    assert bench.committed_traffic("cfg2", "numpy", 4096, 512, "k_discrete_rollout_pipe<>")[0] is None      # another batch
    t5, src5 = bench.committed_traffic("cfg5", "philox", 65536, 512, "k_continuous_rollout_fast<D=12,...>")
    assert src5 == "r03_traffic_cfg5_philox.json" and abs(t5 / (65536 * 512) - 104.8) < 1.0


def test_self_launch_spawns_a_child_torchrun(monkeypatch):
    calls = {}

    def fake_run(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)
    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    args = types.SimpleNamespace(gpus=4)
    try:
        bench.self_launch(args, ["--gpus", "4", "--steps", "20", "--warmup", "5"])
    except SystemExit as e:
        assert e.code == 0
    cmd = calls["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_live_traffic_gives_up_without_profiler(monkeypatch):
    """No rocprofv3 (or no GPU under it) -> None, and the bench line falls back to the committed record."""
    import shutil
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert bench.live_traffic("cfg2", "numpy", 65536, 512) is None


def test_live_traffic_is_skipped_inside_a_profiled_run(monkeypatch):
    """Under `rocprofv3 ... -- python bench.py` no child `rocprofv3 --pmc` is started (it would inherit the tracing
    environment): the line then carries the committed record."""
    import shutil
    import subprocess
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not spawn")))
    monkeypatch.setenv("ROCPROFILER_OUTPUT_PATH", "/tmp/x")
    assert bench.live_traffic("cfg2", "numpy", 65536, 512) is None
    monkeypatch.delenv("ROCPROFILER_OUTPUT_PATH")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.live_traffic("cfg2", "numpy", 65536, 512) is None
