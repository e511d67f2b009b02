"""The product path has no CPU fallback: without the HIP library or without a ROCm device it
raises, it never routes through the oracle or any other CPU code."""
import ast
import os

import pytest
import torch

from mdp_playground_amd import _capi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libmdpp_hip.so")
    with pytest.raises(_capi.MdppError):
        _capi.load()


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine without a GPU")
def test_no_gpu_raises():
    from mdp_playground_amd import RLToyVectorEnv
    with pytest.raises(_capi.MdppError):
        RLToyVectorEnv(num_envs=4, state_space_type="discrete", action_space_type="discrete",
                       state_space_size=8, action_space_size=8, seed=0)


def test_product_never_imports_oracle():
    """No module of the shipped package imports anything under oracle/ (or tools/)."""
    pkg = os.path.join(ROOT, "mdp_playground_amd")
    for fn in os.listdir(pkg):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.startswith("oracle") and not n.startswith("tools"), (fn, n)
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        if fn.endswith((".hip", ".hpp")):
            assert "oracle/" not in open(os.path.join(pkg, "csrc", fn)).read(), fn
