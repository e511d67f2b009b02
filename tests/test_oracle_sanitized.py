"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("race detection / sanitizers").

Every parity verdict rests on oracle/mdpp_oracle.c + np_random.c -- 1 300 lines of pointer arithmetic over caller-sized
buffers.  GPU sanitizers are not available on the pool, the CPU build is where a sanitizer belongs: this test compiles the
two files with -fsanitize=address,undefined -fno-sanitize-recover=all (oracle.build(sanitize=True)) and runs the golden
suites of the oracle (discrete, irrelevant features, grid, continuous, statistics, polygon / continuous images, the
post-processor, numpy's generator) through that build in a child process with libasan preloaded.  Any out-of-bounds
access, use after free, signed overflow, misaligned access or invalid shift aborts the child.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_oracle_golden.py", "tests/test_image_oracle.py", "tests/test_post_oracle_golden.py",
          "tests/test_np_random.py", "tests/test_reference_kats.py"]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_golden_suites_clean_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan on this machine")
    sys.path.insert(0, ROOT)
    from oracle import oracle as ora
    so = ora.build(sanitize=True)
    assert os.path.exists(so)
    # the build is instrumented (an uninstrumented library would make this test vacuous)
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert "__asan_report_load" in syms or "__asan_init" in syms, "the sanitized build carries no ASan instrumentation"
    assert "__ubsan_handle" in syms, "the sanitized build carries no UBSan instrumentation"
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, MDPP_ORACLE_SANITIZE="1", PYTHONPATH=ROOT,
               # CPython itself is not leak-clean; everything else stays on and aborts
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, "sanitized oracle run failed:\n" + tail
    assert " passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_sanitized_build_catches_an_out_of_bounds_write():
    """The harness itself: a deliberate one-past-the-end write through the sanitized library's own np_* entry point must
    abort the child (so a green run above means something)."""
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan on this machine")
    sys.path.insert(0, ROOT)
    from oracle import oracle as ora
    ora.build(sanitize=True)
    code = (
        "import ctypes as C, numpy as np\n"
        "from oracle import oracle as ora\n"
        "L = ora.lib()\n"
        "libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]\n"
        "buf = libc.malloc(32)\n"                         # room for 4 normals, ask for 64: a heap overflow inside the C
        "L.np_philox_normals(1, 0, 0, 0, 8, 8, C.c_void_p(buf))\n"
        "print('SURVIVED')\n")
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, MDPP_ORACLE_SANITIZE="1", PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=23")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "SURVIVED" not in r.stdout and r.returncode != 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "AddressSanitizer" in r.stderr, r.stderr[-1500:]
