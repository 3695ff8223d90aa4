"""The C-ABI shared library loads and exports every symbol include/v2a.h declares (no compute without a GPU); the product path
refuses to run without the HIP extension / on CPU tensors."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "v2a.h")).read()
    return sorted(set(re.findall(r"\b(v2a_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from v2a_hip import _lib
    syms = header_symbols()
    assert len(syms) >= 40
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/v2a.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_host_only_entry_points():
    from v2a_hip._lib import lib
    assert lib.v2a_opt_chunk_elems() > 0 and lib.v2a_opt_state_bytes() > 64
    assert lib.v2a_conv2d_workspace_bytes(256, 1024, 5120) > 0          # small-M / deep-K -> split-K slabs
    assert lib.v2a_conv2d_workspace_bytes(1 << 20, 128, 1152) == 0
    assert lib.v2a_groupnorm_workspace_bytes(2, 16, 256, 8) == 0 and lib.v2a_groupnorm_workspace_bytes(2, 7 * 128 * 128, 128, 32) > 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from v2a_hip import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.V2AError):
        _lib._load()
