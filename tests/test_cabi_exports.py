"""CPU tests: the C-ABI library loads and exports every symbol include/unetk.h declares."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(REPO, "include", "unetk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(unetk_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from unet_amd import _lib
    assert declared_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    from unet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("unet_amd/libunetk.so missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    _lib.load()
    assert _lib.load().unetk_version() >= 100
    assert _lib.load().unetk_conv_pick_bn(768) == 96


def test_missing_library_fails_loudly(monkeypatch):
    from unet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libunetk.so")
    with pytest.raises(ImportError):
        _lib.load()
