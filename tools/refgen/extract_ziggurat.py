#!/usr/bin/env python3
"""Extract numpy's 256-layer ziggurat tables for random_standard_normal().

The algorithm lives in a third-party dependency of the reference (numpy's
random/src/distributions/distributions.c + ziggurat_constants.h, numpy 2.2.6 in
this image); it is not under /root/reference and there is no network, so the
three constant tables (ki_double, wi_double, fi_double: data, not code) are read
out of the installed numpy binary and written as plain .inc initialiser lists.
They are validated by tests/test_np_random.py, which checks millions of normals
produced with these tables against numpy itself.

    python tools/refgen/extract_ziggurat.py
"""
import os
import struct
import numpy

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
DESTS = ("oracle/np_ziggurat_tables.inc", "mdp_playground_amd/csrc/np_ziggurat_tables.inc")
HDR = ("/* numpy %s ziggurat tables for standard_normal (data extracted by\n"
       " * tools/refgen/extract_ziggurat.py; do not edit). */\n")


def tables():
    """(ki, wi, fi) read out of the installed numpy's _generator extension."""
    import glob
    so = glob.glob(os.path.join(os.path.dirname(numpy.__file__), "random", "_generator.*.so"))[0]
    blob = open(so, "rb").read()
    w0 = struct.pack("<d", 8.68362706080130616677e-16)   # wi_double[0]
    k0 = struct.pack("<Q", 0x000EF33D8025EF6A)           # ki_double[0]
    w_off = blob.find(w0)
    k_off = blob.find(k0)
    assert w_off > 0 and k_off == w_off + 2048, (w_off, k_off)
    f_off = w_off - 2048
    fi = struct.unpack_from("<256d", blob, f_off)
    wi = struct.unpack_from("<256d", blob, w_off)
    ki = struct.unpack_from("<256Q", blob, k_off)
    assert fi[0] == 1.0 and all(fi[i] > fi[i + 1] for i in range(255))
    return ki, wi, fi


def render(ki, wi, fi, version):
    """The .inc text: ONE rendering, written to both places (tests/test_np_random.py checks they are this text)."""
    return (HDR % version + "#define NPZ_KI_INIT { \\\n" + ", \\\n".join("  0x%016XULL" % v for v in ki) +
            " }\n#define NPZ_WI_INIT { \\\n" + ", \\\n".join("  %s" % float.hex(v) for v in wi) +
            " }\n#define NPZ_FI_INIT { \\\n" + ", \\\n".join("  %s" % float.hex(v) for v in fi) + " }\n")


if __name__ == "__main__":
    text = render(*tables(), numpy.__version__)
    for dst in DESTS:
        with open(os.path.join(ROOT, dst), "w") as f:
            f.write(text)
        print("wrote", dst)
