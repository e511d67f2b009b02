"""A process that never touches the GPU and starts other programs on request (one JSON line in, one out).
tests/conftest.py starts it BEFORE the pytest process initialises the GPU runtime, so that GPU tests which need a
fresh process (an RCCL rank) get one that is not the fork + exec of a GPU-initialised process."""
import json
import os
import subprocess
import sys

for line in sys.stdin:
    try:
        req = json.loads(line)
        r = subprocess.run(req["argv"], env=dict(os.environ, **req.get("env", {})), cwd=req.get("cwd"),
                           capture_output=True, text=True, timeout=req.get("timeout", 300))
        out = {"returncode": r.returncode, "stdout": r.stdout[-20000:], "stderr": r.stderr[-20000:]}
    except Exception as e:                  # timeouts included
        out = {"returncode": -999, "stdout": "", "stderr": repr(e)}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
