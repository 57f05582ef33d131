"""CPU-side checks of the C-ABI boundary: the library loads, exports every entry that
include/kx_hip.h declares, and the Python binding table names the same set.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

import kx

HEADER = os.path.join(ROOT, "include", "kx_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "kx_gemv_w8_f32" in syms and "kx_mha_decode_f32" in syms and len(syms) > 40


def test_library_exports_every_declared_symbol():
    assert os.path.exists(kx.LIB_PATH), "libkx_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(kx.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_table_matches_header():
    assert sorted(kx.SIGNATURES) == declared_symbols()


def test_error_strings_without_gpu():
    L = kx.lib()
    assert L.kx_error_string(0) == b"ok"
    assert b"invalid" in L.kx_error_string(-1)
    assert b"unsupported" in L.kx_error_string(-2)
    assert L.kx_attn_workspace_bytes(32, 128) == 16384 + 32 * 128 * 132 * 4


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(kx, "_lib", None)
    monkeypatch.setattr(kx, "LIB_PATH", "/nonexistent/libkx_hip.so")
    with pytest.raises(kx.KxError):
        kx.lib()
