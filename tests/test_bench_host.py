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
    monkeypatch.setattr(bench, "_run_group", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not spawn")))
    monkeypatch.setenv("ROCPROFILER_OUTPUT_PATH", "/tmp/x")
    assert bench.live_traffic("cfg2", "numpy", 65536, 512) is None
    monkeypatch.delenv("ROCPROFILER_OUTPUT_PATH")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.live_traffic("cfg2", "numpy", 65536, 512) is None


def test_live_traffic_all_splits_counter_rows_at_the_marker_dispatches(monkeypatch, tmp_path):
    """bench.live_traffic_all: the two child rocprofv3 passes run every workload in ONE process with a marker dispatch
    (k_philox_normals) after each; the counter rows, read in dispatch order, are split into one segment per workload,
    mdpp:: kernels only, reset kernels excluded, FETCH_SIZE counted twice (gfx950), KB -> bytes per launch."""
    import shutil
    import subprocess
    import types
    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3")
    for k in list(__import__("os").environ):
        if k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    specs = [("cfg2", "numpy", 65536, 512), ("cfg3", "numpy", 65536, 512)]
    launches = 3

    def fake_run(cmd, **kw):
        counter = cmd[cmd.index("--pmc") + 1]
        out = cmd[cmd.index("-d") + 1]
        d = __import__("os").path.join(out, "host", "1234")
        __import__("os").makedirs(d)
        rows, did = [], 0

        def row(name, val):
            nonlocal did
            did += 1
            rows.append(f'{did},"{name}",{counter},{val}')
        per = {"FETCH_SIZE": (100.0, 1000.0), "WRITE_SIZE": (400.0, 2000.0), "SQ_INSTS_VALU": (6.0e7, 9.0e8)}[counter]
        for w in range(sum(1 for c in cmd if c.count(":") == 3)):                                  # one segment per workload spec
            row("void mdpp::k_discrete_reset<false>(mdpp::DiscreteArgs)", 5.0)              # constructor reset: excluded
            row("void at::native::vectorized_elementwise_kernel<4>(int)", 77.0)             # torch kernel: excluded
            for _ in range(launches):
                row(("void mdpp::k_discrete_rollout_lean<true>(mdpp::DiscreteArgs, int)", "void mdpp::k_continuous_rollout_fast<12, 1>(mdpp::ContinuousArgs)")[w], per[w])
            row("mdpp::k_philox_normals(unsigned long, long)", 0.0)                         # the marker
        with open(__import__("os").path.join(d, "1234_counter_collection.csv"), "w") as f:
            f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n" + "\n".join(reversed(rows)) + "\n")   # (file order != dispatch order)
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(bench, "_run_group", lambda cmd, timeout, **kw: fake_run(cmd, **kw).returncode)   # (the child rocprofv3 passes)
    res = bench.live_traffic_all(specs, launches=launches)
    assert set(res) == {"cfg2", "cfg3"}
    assert res["cfg2"]["bytes_per_launch"] == int((2 * 100.0 + 400.0) * 1024)
    assert res["cfg3"]["bytes_per_launch"] == int((2 * 1000.0 + 2000.0) * 1024)
    assert "k_discrete_rollout_lean" in res["cfg2"]["kernels"] and "reset" not in res["cfg2"]["kernels"]
    # round 5: the third pass, vector instructions issued per launch -> valu_frac (a failure of that pass alone only drops it)
    assert res["cfg2"]["valu_insts_per_launch"] == 6.0e7 and res["cfg3"]["valu_insts_per_launch"] == 9.0e8
    v = bench.valu_roofline(res["cfg3"]["valu_insts_per_launch"], 600.0, 0.70)
    assert abs(v["valu_frac"] - 9.0e8 / 1024 * 4 / 2.4e9 * 1e6 / 600.0) < 1e-12 and v["bound"] == "valu"
    assert bench.valu_roofline(None, 600.0, 0.70) == {"valu_frac": None, "bound": "hbm"}
    real = fake_run

    def fail_third(cmd, **kw):
        if "SQ_INSTS_VALU" in cmd:
            return types.SimpleNamespace(returncode=1)
        return real(cmd, **kw)
    monkeypatch.setattr(bench, "_run_group", lambda cmd, timeout, **kw: fail_third(cmd, **kw).returncode)
    res2 = bench.live_traffic_all(specs, launches=launches)
    assert res2["cfg2"]["bytes_per_launch"] == res["cfg2"]["bytes_per_launch"] and res2["cfg2"]["valu_insts_per_launch"] is None
    one = bench.live_traffic("cfg2", "numpy", 65536, 512, launches=launches)
    assert one[0] == res["cfg2"]["bytes_per_launch"]


def test_action_rotation_is_larger_than_the_infinity_cache():
    """>= 4 distinct tensors, >= 512 MiB together, capped at 32 tensors (sizes only: the tensors need a device)."""
    import torch
    made = []

    def fake_make(wl, K, N, device, seed):
        made.append(seed)
        return torch.empty(0, dtype=torch.int32).new_empty((K, N), device="meta")
    import pytest
    mp = pytest.MonkeyPatch()
    mp.setattr(bench, "make_actions", fake_make)
    try:
        a = bench.action_rotation(bench.WORKLOADS["cfg2"], 512, 65536, "meta", 7)
        assert len(a) == 4 and len(set(made)) == 4 and sum(x.numel() * 4 for x in a) >= 512 << 20
        made.clear()
        a = bench.action_rotation(bench.WORKLOADS["cfg4"], 512, 8192, "meta", 7)
        assert len(a) == 32 and len(set(made)) == 32
    finally:
        mp.undo()
    assert bench.leg_name("cfg5", "philox") == "cfg5_philox" and bench.leg_name("cfg2", "numpy") == "cfg2"
