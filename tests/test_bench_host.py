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


def _canned_full_record():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_driver_argv.json")) as f:
        return json.load(f)        # round 5's own 26 KB line: the one the driver could not parse


def test_contract_line_is_compact_and_complete():
    """BENCH_r05.json `parsed: null`: the one JSON line had grown to 26 KB.  The contract line is now built from the full
    record by bench.compact_line: every contract key, `roofline` and `cpu_baseline`, <= 4 KiB -- also when the record holds
    many more workload legs than today's."""
    import json
    full = _canned_full_record()
    assert len(json.dumps(full)) > 20000
    s = bench.compact_line(full)
    assert len(s.encode()) <= bench.LINE_MAX_BYTES == 4096 and "\n" not in s
    line = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["envs_per_gpu"] == 65536 and line["config"]["fuse"] == 512 and len(line["config"]["workload"]) <= 200
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["frac"] - full["roofline"]["frac"]) < 1e-3
    assert r["traffic"] == full["roofline"]["traffic"] and r["kernel"] == full["roofline"]["kernel"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and abs(c["value"] / full["cpu_baseline"]["value"] - 1) < 1e-4 and c["sample"]
    assert abs(line["value"] / full["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / full["ms_per_step"] - 1) < 1e-4
    assert set(line["workloads"]) == set(full["workloads"])
    w = line["workloads"]["img100_all"]
    assert len(w) == 5 and abs(w[1] - full["workloads"]["img100_all"]["frac"]) < 1e-3 and 1.2 < w[3] < 1.35
    # four times the legs: optional parts are dropped, the contract keys never
    big = dict(full, workloads={f"{k}_{i}": v for i in range(8) for k, v in full["workloads"].items()})
    s2 = bench.compact_line(big)
    line2 = json.loads(s2)
    assert len(s2.encode()) <= 4096 and "roofline" in line2 and "cpu_baseline" in line2 and line2["value"] == line["value"]


def test_contract_line_is_the_last_line_of_stdout(capfd, tmp_path):
    """emit(): the full record goes to a file, C stdio is flushed FIRST (RCCL's version banner sits in printf's buffer until
    exit and used to land after the JSON), and the contract line is the last thing on stdout."""
    import ctypes
    import json
    libc = ctypes.CDLL(None)
    libc.printf(b"RCCL version : 2.26.6-HEAD:64f48b6\nHostname     : runc\n")     # buffered by C stdio (stdout is not a tty here)
    full = _canned_full_record()
    out = tmp_path / "sub" / "bench_detail.json"
    bench.emit(full, str(out))
    libc.fflush(None)
    cap = capfd.readouterr().out
    lines = [ln for ln in cap.splitlines() if ln.strip()]
    last = json.loads(lines[-1])
    assert last["metric"] == "env-steps/sec" and "roofline" in last and "cpu_baseline" in last and len(lines[-1]) <= 4096
    assert any(ln.startswith("RCCL version") for ln in lines[:-1])
    assert sum(1 for ln in lines if ln.startswith("{")) == 1              # ONE JSON line
    detail = json.loads(out.read_text())
    assert detail["workloads"]["cfg4"]["launch_us_runs"] and last["detail"] == str(out)
