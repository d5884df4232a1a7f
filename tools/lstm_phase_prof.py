#!/usr/bin/env python3
"""Phase cycles of one backward step of the hidden-128 training recurrence (bilstm4_bwd_kernel), debug build only:
    a libamtx.so whose lstm.o was compiled with -DAMTX_LSTM_TIMING, loaded through AMTX_LIB_PATH; python tools/lstm_phase_prof.py [clips=8]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amt_tools_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, H = 625, 128
L = _lib.lib()
prof = L.amtxdbg_lstm_prof
prof.restype = C.c_int; prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
whf = torch.randn(4 * H, H, device=dev, generator=g) * 0.05
whb = torch.randn(4 * H, H, device=dev, generator=g) * 0.05
n = int(L.amtx_bilstm_packed_elems(2))
ff = torch.empty(n, dtype=torch.int16, device=dev); fb = torch.empty(n, dtype=torch.int16, device=dev)
st = _lib.current_stream(torch.device(dev))
_lib.check(L.amtx_bilstm_pack_device(_lib.ptr(whf), _lib.ptr(whb), 2, _lib.ptr(ff), _lib.ptr(fb), st))
xproj = torch.randn(B, T, 2, 4 * H, device=dev, generator=g) * 0.5
out = torch.empty(B, T, 2 * H, device=dev); save = torch.empty(B, T, 2, 5, H, device=dev)
_lib.check(L.amtx_bilstm_train_fwd(_lib.ptr(xproj), _lib.ptr(ff), 2, _lib.ptr(out), _lib.ptr(save), B, T, st))
dout = torch.randn(B, T, 2 * H, device=dev, generator=g) * 0.1
dx = torch.empty(B, T, 2, 4 * H, device=dev)
buf = (C.c_ulonglong * 4)()
for rep in range(2):
    torch.cuda.synchronize(); prof(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.amtx_bilstm_train_bwd(_lib.ptr(dout), _lib.ptr(save), _lib.ptr(fb), 2, _lib.ptr(dx), B, T, st))
    e1.record(); torch.cuda.synchronize(); prof(buf, 0)
v = list(buf)
print(f'{B} clips x {T} steps: {e0.elapsed_time(e1) * 1e3:.0f} us; wave 0 of block 0, cycles per step: elementwise + LDS writes {v[0] / v[3]:.0f}, '
      f'barrier {v[1] / v[3]:.0f}, fragment reads + MFMAs {v[2] / v[3]:.0f} (total {(v[0] + v[1] + v[2]) / v[3]:.0f})')
