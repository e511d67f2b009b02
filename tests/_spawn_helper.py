"""A process that never touches the GPU and starts other programs on request (one JSON line in, one out).
tests/conftest.py starts it BEFORE the pytest process initialises the GPU runtime, so that GPU tests which need a
fresh process (an RCCL rank) get one that is not the fork + exec of a GPU-initialised process."""
import json
import os
import signal
import subprocess
import sys

for line in sys.stdin:
    try:
        req = json.loads(line)
        # its own session: on a timeout the WHOLE group goes (a torch.distributed.run child has rank grandchildren,
        # bench.py a rocprofv3 one -- left alone they would keep the box's one GPU busy under the later tests)
        p = subprocess.Popen(req["argv"], env=dict(os.environ, **req.get("env", {})), cwd=req.get("cwd"),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = p.communicate(timeout=req.get("timeout", 300))
            out = {"returncode": p.returncode, "stdout": so[-20000:], "stderr": se[-20000:]}
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
            so, se = p.communicate()
            out = {"returncode": -999, "stdout": (so or "")[-20000:], "stderr": "timeout; process group killed\n" + (se or "")[-20000:]}
    except Exception as e:
        out = {"returncode": -999, "stdout": "", "stderr": repr(e)}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
