import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_helper = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runs: a helper process that never touches the GPU, started before this process does; tests that need a
    # fresh process (an RCCL rank) ask it to start one (tests/_spawn_helper.py)
    global _helper
    expr = (config.getoption("-m") or "").strip()
    if "gpu" in expr and "not gpu" not in expr:
        import subprocess
        _helper = subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "_spawn_helper.py")],
                                   stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT)


def pytest_unconfigure(config):
    global _helper
    if _helper is not None:
        try:
            _helper.stdin.close()
            _helper.wait(timeout=10)
        except Exception:
            _helper.kill()
        _helper = None


@pytest.fixture
def spawn_fresh():
    """spawn_fresh(argv, env={}, timeout=300) -> {"returncode", "stdout", "stderr"} of a program run in a fresh
    process whose parent never initialised the GPU."""
    import json
    if _helper is None:
        pytest.skip("the spawn helper runs only under -m gpu")

    def run(argv, env=None, timeout=300):
        _helper.stdin.write(json.dumps({"argv": argv, "env": env or {}, "timeout": timeout, "cwd": ROOT}) + "\n")
        _helper.stdin.flush()
        line = _helper.stdout.readline()
        if not line:
            raise RuntimeError("the spawn helper died")
        return json.loads(line)
    return run


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
