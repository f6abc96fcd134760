"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/dsea.h declares, and its
argument validation (which runs before any device work) behaves.  No compute calls (no GPU here)."""
import ctypes
import os
import re
from ctypes import byref, c_size_t, c_void_p

import pytest

from conftest import ROOT
from dominantsparseeigenad_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dsea.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsea_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 29
    for name in names:
        assert hasattr(lib, name), "libdsea.so does not export %s" % name
    # and the ctypes table covers the header one to one
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.dsea_version() >= 100
    assert lib.dsea_error_string(0) == b"ok"
    assert b"aligned" in lib.dsea_error_string(-2)


def test_argument_validation_without_device():
    lib = _lib.load()
    nbytes = c_size_t()
    assert lib.dsea_ws_bytes(1 << 20, 200, byref(nbytes)) == 0
    # partial sums (8192 wave tiles x k) + 4 work vectors
    assert nbytes.value >= 8192 * 200 * 8 + 4 * (1 << 20) * 8
    assert lib.dsea_ws_bytes(0, 10, byref(nbytes)) == -1
    h = c_void_p()
    assert lib.dsea_ws_create(None, 0, 10, 10, byref(h)) == -1
    assert lib.dsea_op_create_tfim(0, 0, 0, None, 1.0, 1.0, byref(h)) == -1          # L < 1
    assert lib.dsea_op_create_tfim(10, 11, 0, None, 1.0, 1.0, byref(h)) == -1        # L_local > L
    assert lib.dsea_op_create_tfim(10, 8, 100, None, 1.0, 1.0, byref(h)) == -1       # slab not aligned
    assert lib.dsea_op_create_tfim(10, 8, 256, None, 1.0, 1.0, byref(h)) == 0
    n = ctypes.c_int64()
    assert lib.dsea_op_dim(h, byref(n)) == 0 and n.value == 256
    assert lib.dsea_op_destroy(h) == 0
    assert lib.dsea_op_create_csr(10, 5, None, None, None, byref(h)) == -1
    assert lib.dsea_dot(None, None, None, 10, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdsea.so")
    with pytest.raises(_lib.DseaError):
        _lib.load()


def test_integration_md_indexes_every_entry_point():
    """INTEGRATION.md section 4 = tools/abi_index.py on the header as it stands: one row per declared function, and the
    declared functions are the library's exports (test_library_exports_every_declared_symbol)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("abi_index", os.path.join(ROOT, "tools", "abi_index.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    block, count = mod.block()
    assert count == len(header_symbols())
    assert sorted(r[0] for r in mod.rows()) == sorted(header_symbols())
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert block in doc, "INTEGRATION.md section 4 is stale: python tools/abi_index.py --write"
