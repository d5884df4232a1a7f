#!/usr/bin/env python3
"""Single-clip latency (the reference's run_offline pattern: one track per call): audio resident on the device -> piano rolls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip
model, mel, sd = bench.build_model('cuda:0', 'bf16')
for B in (1, 2, 4, 8, 16, 32):
    audio = torch.from_numpy(np.stack([synth_clip(i) for i in range(B)])).cuda()
    with torch.no_grad():
        for _ in range(3): model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f'{B} clip(s) x 625 frames per call: {dt * 1e3:.2f} ms per call, {B * 625 / dt / 1e6:.2f} M frames/s', flush=True)
