"""CPU tests: the C-ABI library loads and exports every symbol include/az_engine.h declares;
the Python binding's signature table covers exactly those symbols.  No compute calls."""
import ctypes
import os
import re

import pytest

from alphazero_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libazamd.so not built: run __graft_entry__.build()")
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_struct_sizes_match_header():
    # az_state_blob: 8 boards of 384 bytes + 14 int32
    assert ctypes.sizeof(_lib.AzStateBlob) == 8 * 384 + 14 * 4
    assert ctypes.sizeof(_lib.AzCounters) == 21 * 8


def test_engine_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alphazero_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine("go", 9, 9)
