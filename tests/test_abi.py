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
    assert ctypes.sizeof(_lib.AzCounters) == 25 * 8


def header_struct_fields(name):
    """Member names of `typedef struct <name> {...}` in include/az_engine.h, in order."""
    txt = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]          # drop the type
        fields += [re.sub(r"\[.*", "", n).strip() for n in names.split(",")]
    return fields


def test_library_reports_the_struct_sizes_of_the_binding():
    L = _lib.lib()
    sizes = (ctypes.c_int32 * 3)()
    L.az_abi_sizes(sizes)
    assert tuple(sizes) == (ctypes.sizeof(_lib.AzConfig), ctypes.sizeof(_lib.AzStateBlob), ctypes.sizeof(_lib.AzCounters))


def test_binding_structs_follow_the_header_member_by_member():
    for cls, name in ((_lib.AzConfig, "az_config"), (_lib.AzStateBlob, "az_state_blob"), (_lib.AzCounters, "az_counters")):
        assert [f[0] for f in cls._fields_] == header_struct_fields(name), name


def test_integration_guide_stub_matches_the_binding():
    """The ctypes stub printed in the integration guide is what a maintainer would copy: same members, same order."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = txt[txt.index("class AzConfig(C.Structure)"):txt.index("L.az_engine_create.restype")]
    assert re.findall(r'\("([a-z0-9_]+)", C\.c_', stub) == [f[0] for f in _lib.AzConfig._fields_]


def test_wrong_struct_size_is_refused():
    """A binding compiled against another header must fail at az_engine_create, before any field is trusted."""
    L = _lib.lib()
    cfg = _lib.AzConfig(struct_size=ctypes.sizeof(_lib.AzConfig) - 4, game=0, board_height=9, board_width=9, n_slots=1)
    assert not L.az_engine_create(ctypes.byref(cfg))
    assert b"struct_size" in L.az_last_error(None)


def test_engine_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alphazero_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine("go", 9, 9)
