#!/usr/bin/env python3
"""Per-step wall times of the bench workload (synchronised after every step) -- for chasing one-off stalls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools, _lib
from amt_tools_amd.synth import synth_clip
model, mel, sd = bench.build_model('cuda:0', 'bf16')
B = 512
base = np.stack([synth_clip(i) for i in range(8)])
audio = torch.from_numpy(base).to('cuda:0').repeat((B + 7) // 8, 1)[:B].contiguous()
batch = {tools.KEY_AUDIO: audio}
ts = []
for i in range(3):
    with torch.no_grad(): out = model.run_on_batch(batch)
L = _lib.lib(); eng = model._get_engine(torch.device('cuda:0'))
_lib.check(L.amtx_of_profile_enable(eng.handle, 1)); mel._prof_events = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(14):
    t0 = time.perf_counter()
    with torch.no_grad(): out = model.run_on_batch(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
print('enqueue/total ms per step:', ' '.join(f'{a:.1f}/{b:.1f}' for a, b in ts))
