import torch, time
x = torch.empty(512*319999, dtype=torch.float32).pin_memory()
for _ in range(2): y = x.to('cuda:0', non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): y = x.to('cuda:0', non_blocking=True)
torch.cuda.synchronize()
dt=(time.perf_counter()-t0)/5
print(f'H2D pinned {x.numel()*4/1e9:.2f} GB in {dt*1e3:.1f} ms = {x.numel()*4/dt/1e9:.1f} GB/s')
z = torch.empty(512*88*625*2, dtype=torch.float32, device='cuda:0')
h = torch.empty(z.shape, dtype=torch.float32).pin_memory()
for _ in range(2): h.copy_(z, non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): h.copy_(z, non_blocking=True)
torch.cuda.synchronize()
dt=(time.perf_counter()-t0)/5
print(f'D2H pinned {z.numel()*4/1e9:.2f} GB in {dt*1e3:.1f} ms = {z.numel()*4/dt/1e9:.1f} GB/s')
