"""The C-ABI shared library loads and exports every symbol include/jgrid.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jgrid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(jg):
    L = ctypes.CDLL(jg._lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_torch_or_oracle_in_product():
    """The product never imports the oracle (parity would be void) and the ABI has no torch types."""
    pkg = os.path.join(ROOT, "juliagrid.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "jg_oracle" not in src and "jgo_" not in src, f
                if f.endswith((".hip", ".cpp", ".hpp")):
                    assert "torch" not in src, f


def test_missing_library_fails_loudly(jg, monkeypatch):
    import pytest
    monkeypatch.setattr(jg._lib, "_lib", None)
    monkeypatch.setattr(jg._lib, "LIB_PATH", "/nonexistent/libjgrid_hip.so")
    with pytest.raises(ImportError):
        jg._lib.lib()
