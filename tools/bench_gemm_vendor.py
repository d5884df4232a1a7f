#!/usr/bin/env python3
"""What the vendor library (hipBLASLt through torch.matmul, bf16) reaches on the engine's GEMM shapes -- a yardstick for
amtx_linear_fwd (tools/bench_gemm.py), not part of the product path."""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 320000
# fc1, the two input projections, the folded pitch head (N = 88 runs as a 128-wide tile); DESIGN.md quotes M = 320000 and M = 640000
for (n, k) in [(512, 3648), (1024, 512), (1024, 192), (128, 3648)]:
    a = torch.randn(M, k, device='cuda').bfloat16()
    w = (torch.randn(n, k, device='cuda') / k ** 0.5).bfloat16()
    for _ in range(3): c = a @ w.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'hipBLASLt M={M} N={n} K={k}: {ms:.3f} ms  {2.0 * M * n * k / ms / 1e9:.0f} TFLOP/s  {M * k * 2 / ms / 1e6:.0f} GB/s of A', flush=True)
