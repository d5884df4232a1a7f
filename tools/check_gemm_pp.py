#!/usr/bin/env python3
"""Race screen + A/B timing of the two-group ring GEMM (AMTX_GEMM_PP=1) against the default kernels: same operands, the k order of
the accumulation is the same, so the C tiles must be bit-identical; repeated runs must agree with themselves."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child():
    import numpy as np, torch
    from amt_tools_amd import _lib
    L = _lib.lib(); s = _lib.current_stream()
    out = {}
    for (m, n, k) in [(70000, 512, 3648), (70001, 1024, 512), (1000, 256, 128), (320000, 512, 3648)]:
        g = torch.Generator().manual_seed(m + n + k)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).numpy()
        packed = np.zeros(L.amtx_linear_packed_elems(n, k, 1), dtype=np.uint16)
        _lib.check(L.amtx_linear_pack(_lib.ptr(w), n, k, 1, _lib.ptr(packed)))
        wp = torch.from_numpy(packed.view(np.int16)).cuda()
        a = torch.randn(m, k, device='cuda', generator=torch.Generator('cuda').manual_seed(1)).bfloat16()
        bias = torch.randn(n, generator=g).cuda()
        c = torch.zeros(m, n, dtype=torch.bfloat16, device='cuda')
        def run():
            _lib.check(L.amtx_linear_fwd(_lib.ptr(a), k, 0, _lib.ptr(wp), 1, _lib.ptr(bias), _lib.ptr(c), n, 0, m, n, k, s))
        run(); torch.cuda.synchronize()
        first = c.clone()
        same = True
        for _ in range(6):
            c.zero_(); run(); torch.cuda.synchronize()
            same = same and bool(torch.equal(c, first))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ref = (a[:2048].float() @ torch.from_numpy(w).cuda().bfloat16().float().t() + bias)
        err = (first[:2048].float() - ref).abs().max().item() / ref.abs().max().item()
        out[(m, n, k)] = first.view(torch.int16).cpu()
        print(f'  M={m} N={n} K={k}: {ms:.3f} ms {2.0 * m * n * k / ms / 1e9:.0f} TFLOP/s  self-consistent={same}  rel err vs torch {err:.2e}', flush=True)
    torch.save(out, sys.argv[2])

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    child(); sys.exit(0)
import torch
env = dict(os.environ)
env.pop('AMTX_GEMM_PP', None)
env['AMTX_GEMM_NO_PP'] = '1'
print('default kernels'); subprocess.run([sys.executable, __file__, 'child', '/tmp/gemm_ref.pt'], env=env, check=True)
env['AMTX_GEMM_PP'] = '1'; env.pop('AMTX_GEMM_NO_PP')
print('AMTX_GEMM_PP=1'); subprocess.run([sys.executable, __file__, 'child', '/tmp/gemm_pp.pt'], env=env, check=True)
a, b = torch.load('/tmp/gemm_ref.pt'), torch.load('/tmp/gemm_pp.pt')
for k in a: print(k, 'bit-identical' if torch.equal(a[k], b[k]) else f'DIFFERENT in {(a[k] != b[k]).sum().item()} elements')
