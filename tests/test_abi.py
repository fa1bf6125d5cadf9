"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/tbk.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from tiebrush_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tbk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tbk_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_match_binding_list():
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.load()
    assert L.tbk_abi_version() == 8
    for s in _declared_symbols():
        assert hasattr(L, s), s
    assert L.tbk_strerror(-7).decode().startswith("unknown opcode")


def test_create_fails_loudly_without_gpu():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    h = C.c_void_p()
    assert L.tbk_create(0, C.byref(h)) == -9  # TBK_ENODEVICE, never a silent CPU path


def test_host_library_exports_every_symbol_of_its_header():
    txt = open(os.path.join(ROOT, "include", "tbh_host.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tbh_[a-z0-9_]+)\s*\(", txt)))
    assert declared == sorted(_lib.HOST_SYMBOLS)
    H = _lib.load_host()
    assert H.tbh_abi_version() == 1
    for s in declared:
        assert hasattr(H, s), s
