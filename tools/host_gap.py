#!/usr/bin/env python3
"""Back-to-back (unsynchronised) steps of the bench workload under the four combinations of per-kernel event timing
(engine HIP events / front-end torch events): enqueue time vs total time per step."""
import sys, os, time
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
def step():
    with torch.no_grad():
        return model.run_on_batch(batch)
for _ in range(3): out = step()
torch.cuda.synchronize()
L = _lib.lib(); eng = model._get_engine(torch.device('cuda:0'))
for eng_prof, mel_prof in ((1, 1), (0, 0), (1, 0), (0, 1), (1, 1)):
    _lib.check(L.amtx_of_profile_enable(eng.handle, eng_prof))
    mel._prof_events = [] if mel_prof else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'engine events {eng_prof} mel events {mel_prof}: enqueue {1e2*(t1-t0):.2f} ms/step, total {1e2*(t2-t0):.2f} ms/step', flush=True)
