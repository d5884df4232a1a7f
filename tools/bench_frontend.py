#!/usr/bin/env python3
"""Micro-benchmark of the front-end kernels (K1 power, K2 scale) with torch CUDA events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from amt_tools_amd.features import MelSpec
from amt_tools_amd.synth import synth_clip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = 319999
base = np.stack([synth_clip(i) for i in range(4)])
x = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
mod = MelSpec(sample_rate=22050)
for _ in range(3):
    power, cmax = mod.power_batch(x)
    out = mod.scale_batch(power, cmax, model_layout=True)
torch.cuda.synchronize()
T = power.shape[1]
for name, fn in (('K1 power', lambda: mod.power_batch(x)), ('K2 scale', lambda: mod.scale_batch(power, cmax, model_layout=True))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    frames = B * T
    print(f'{name}: {ms:.3f} ms  {frames / ms * 1e3 / 1e6:.1f} Mframes/s  algorithmic GB/s = {frames * (2048 + 916) / ms / 1e6:.1f}')
