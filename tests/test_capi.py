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
    assert L.amtx_bilstm_h_packed_elems(256, 1) == 2 * 1024 * 256 and L.amtx_bilstm_h_packed_elems(128, 2) == L.amtx_bilstm_packed_elems(2)
    assert L.amtx_conv3x3g_packed_elems(48, 96, 1) == 6 * 14 * 512 and L.amtx_conv3x3g_packed_elems(32, 40, 1) == 0


def test_errors_are_reported_not_swallowed():
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.amtx_of_model_create(C.byref(h), 229, 1, 6, 88, 1, 0)    # model_complexity 6: no kernels for its channel counts
    assert rc < 0 and b'model_complexity' in L.amtx_last_error()
    import pytest
    with pytest.raises(_lib.AmtxError):
        _lib.check(rc, 'amtx_of_model_create')
    assert os.path.exists(_lib.LIB_PATH)


def test_missing_extension_fails_loudly_without_a_cpu_fallback(monkeypatch):
    """No HIP extension -> the product path raises; nothing quietly computes on the CPU (oracle/ is never imported by the package)."""
    import sys
    import numpy as np
    import pytest
    from amt_tools_amd.features import MelSpec
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libamtx.so')
    with pytest.raises(_lib.AmtxError, match='no CPU fallback'):
        _lib.lib()
    with pytest.raises(Exception):          # AmtxError (no library) -- never a silently computed spectrogram
        MelSpec(sample_rate=22050).process_audio(np.zeros(4096, dtype=np.float32))
    assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules if 'amt_tools_amd' in str(getattr(sys.modules[m], '__file__', '')))
    import amt_tools_amd, pathlib
    src = ''.join(p.read_text() for p in pathlib.Path(amt_tools_amd.__file__).parent.glob('*.py'))
    assert 'import oracle' not in src and 'from oracle' not in src
