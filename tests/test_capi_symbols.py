"""The C-ABI library loads and exports every entry point include/tlab_amd.h declares, and the ctypes table of tlab_amd/lib.py
lists exactly those (no compute call: runs without a GPU)."""
import ctypes
import os
import re

import tlab_amd
from tlab_amd import lib as tl

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols(which="tlab_amd.h"):
    names = []
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if hdr != which:
            continue
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        names += re.findall(r"\b(tlab_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    assert len(declared_symbols()) > 60


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(tlab_amd.lib_path())
    missing = [n for n in declared_symbols() if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_table_matches_the_header():
    decl = set(declared_symbols())
    table = set(tl.SIGNATURES)
    assert decl - table == set(), "declared in include/tlab_amd.h but not bound in tlab_amd/lib.py: %s" % sorted(decl - table)
    assert table - decl == set(), "bound in tlab_amd/lib.py but not declared in include/tlab_amd.h: %s" % sorted(table - decl)


def test_load_fails_loudly_without_the_library(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(tl, "_LIB", None)
    monkeypatch.setattr(tl, "_HERE", str(tmp_path))
    with pytest.raises(tl.TlabError):
        tl.load()


def test_comm_library_exports_every_declared_symbol_and_the_binding_matches():
    """include/tlab_amd_comm.h (RCCL transpositions) lives in its own library, libtlab_amd_comm.so, bound by tlab_amd/comm.py."""
    from tlab_amd import comm as tc
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["tlab_amd.h", "tlab_amd_comm.h"]
    decl = set(declared_symbols("tlab_amd_comm.h"))
    assert len(decl) >= 13
    L = tc.load()
    assert not [n for n in decl if not hasattr(L, n)]
    assert decl == set(tc.SIGNATURES), (sorted(decl - set(tc.SIGNATURES)), sorted(set(tc.SIGNATURES) - decl))
