#!/usr/bin/env python3
"""Section cycles of the two-group ring GEMM (debug build: AMTX_EXTRA_FLAGS=-DAMTX_GEMM_TIMING, run with AMTX_GEMM_PP=1)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import _lib
L = _lib.lib(); D = ctypes.CDLL(_lib.LIB_PATH); s = _lib.current_stream()
m, n, k = 320000, 512, 3648
w = (np.random.randn(n, k) / k ** 0.5).astype(np.float32)
packed = np.zeros(L.amtx_linear_packed_elems(n, k, 1), dtype=np.uint16)
_lib.check(L.amtx_linear_pack(_lib.ptr(w), n, k, 1, _lib.ptr(packed)))
wp = torch.from_numpy(packed.view(np.int16)).cuda()
a = torch.randn(m, k, device='cuda').bfloat16(); c = torch.empty(m, n, dtype=torch.bfloat16, device='cuda'); bias = torch.zeros(n, device='cuda')
def run(): _lib.check(L.amtx_linear_fwd(_lib.ptr(a), k, 0, _lib.ptr(wp), 1, _lib.ptr(bias), _lib.ptr(c), n, 0, m, n, k, s))
for _ in range(3): run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
D.amtxdbg_gemm_prof(buf, 1)
run(); torch.cuda.synchronize()
D.amtxdbg_gemm_prof(buf, 1)
tiles_per_block = (m // 256 + (1 if m % 256 else 0)) * (n // 256) / 256.0
names = ['L1 reads', 'B1', 'lgkm', 'M1', 'B2', 'vmcnt', 'L2 reads+DMA', 'B1b', 'lgkm b', 'M2', 'epilogue', 'B2b']
for g in range(2):
    st = max(1, buf[g * 16 + 12])
    tot = sum(buf[g * 16 + i] for i in range(12))
    print(f'group {g}: {tot / st:.0f} cycles per stage: ' + ', '.join(f'{nm} {buf[g * 16 + i] / st:.0f}' for i, nm in enumerate(names)) + f'; epilogue-only {buf[g * 16 + 13] / st:.0f} per stage')
