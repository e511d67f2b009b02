"""The C-ABI library loads and exports every symbol include/mdpp.h declares (no compute, no GPU)."""
import ctypes
import os
import re

from mdp_playground_amd import _capi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "mdpp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdpp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 18
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_capi.EXPORTS) == names


def test_abi_version_and_struct_size():
    lib = _capi.load()
    assert lib.mdpp_abi_version() == _capi.MDPP_ABI_VERSION
    # create() validates the version field before touching any GPU state
    cfg = _capi.MdppConfig()
    cfg.abi_version = 99
    h = ctypes.c_void_p()
    assert lib.mdpp_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    assert b"abi_version" in lib.mdpp_last_error(None)
