import torch
n = 2**31  # bytes each
a = torch.empty(n, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
a.fill_(1)
for _ in range(2): b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'copy {n/1e9:.1f} GB: {ms:.3f} ms, {2*n/ms/1e9:.2f} TB/s (read+write)')
x = torch.empty(n // 4, dtype=torch.float32, device='cuda').fill_(1.0)
for _ in range(2): s = x.sum()
e0.record()
for _ in range(10): s = x.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'read-only sum {n/1e9:.1f} GB: {ms:.3f} ms, {n/ms/1e9:.2f} TB/s')
e0.record()
for _ in range(10): x.fill_(2.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'write-only fill {n/1e9:.1f} GB: {ms:.3f} ms, {n/ms/1e9:.2f} TB/s')
