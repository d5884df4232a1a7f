#!/usr/bin/env python3
"""Micro-benchmark of amtx_linear_fwd (bf16 x bf16 direct-to-LDS path) on the shapes of the engine.
Usage: python tools/bench_gemm.py [M]        prints ms and TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import _lib

BF16, F32 = 0, 1
import ctypes
C_DBG = ctypes.CDLL(_lib.LIB_PATH)   # debug builds (-DAMTX_GEMM_TIMING) export amtxdbg_gemm_prof
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 320000
s = _lib.current_stream()
for (n, k) in [(1024, 512), (1024, 192), (512, 3648), (512, 1024), (512, 512), (512, 64), (1024, 64), (256, 512), (128, 512)]:
    w = np.random.randn(n, k).astype(np.float32) / k ** 0.5
    packed = np.zeros(L.amtx_linear_packed_elems(n, k, 1), dtype=np.uint16)
    _lib.check(L.amtx_linear_pack(_lib.ptr(w), n, k, 1, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    a = torch.randn(M, k, device='cuda').bfloat16()
    c = torch.empty(M, n, dtype=torch.bfloat16, device='cuda')
    bias = torch.zeros(n, device='cuda')
    def run():
        _lib.check(L.amtx_linear_fwd(_lib.ptr(a), int(os.environ.get("AMTX_BENCH_LDA0", "1")) * k, BF16, _lib.ptr(wp), 1, _lib.ptr(bias), _lib.ptr(c), n, BF16, M, n, k, s))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (M * k * 2 + M * n * 2) / 1e9
    if hasattr(C_DBG, 'amtxdbg_gemm_prof'):
        buf = (ctypes.c_ulonglong * 8)()
        C_DBG.amtxdbg_gemm_prof(buf, 1)
        run(); torch.cuda.synchronize()
        C_DBG.amtxdbg_gemm_prof(buf, 1)
        tot = sum(buf[i] for i in range(4)) or 1
        print('   k-loop cycles of wave 0 per block: ' + ', '.join(f'{n} {buf[i] / max(1, buf[4]):.0f} ({100 * buf[i] / tot:.0f}%)' for i, n in enumerate(['issue', 'compute', 'vmcnt wait', 'barrier'])))
    print(f'M={M} N={n} K={k}: {ms:.3f} ms  {2.0 * M * n * k / ms / 1e9:.0f} TFLOP/s  {gb / ms * 1e3:.0f} GB/s (A+C)', flush=True)
