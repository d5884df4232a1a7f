"""CPU-side checks of the C ABI: the shared library loads, exports every symbol include/amtx.h declares,
and the ctypes table covers all of them.  No compute calls (no GPU here)."""
import os

from amt_tools_amd import _lib


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f'{name} declared in include/amtx.h but not exported'
    assert set(declared) == set(_lib._SIGNATURES), set(declared) ^ set(_lib._SIGNATURES)
    assert L.amtx_version() >= 100


def test_host_only_queries():
    L = _lib.lib()
    assert L.amtx_linear_packed_elems(88, 176, 1) == 128 * 192
    assert L.amtx_linear_packed_elems(512, 3648, 2) == 2 * 512 * 3648
    assert L.amtx_conv3x3_packed_elems(64, 1) == 9 * 4 * 512
    assert L.amtx_bilstm_packed_elems(2) == 2 * 2 * 512 * 128


def test_errors_are_reported_not_swallowed():
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.amtx_of_model_create(C.byref(h), 229, 1, 3, 88, 1, 0)    # OnsetsFrames2 shape: not implemented yet
    assert rc < 0 and b'model_complexity' in L.amtx_last_error()
    import pytest
    with pytest.raises(_lib.AmtxError):
        _lib.check(rc, 'amtx_of_model_create')
    assert os.path.exists(_lib.LIB_PATH)
