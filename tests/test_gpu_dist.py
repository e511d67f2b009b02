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
