#!/usr/bin/env python3
"""Run the bench workload a few times without timing (target program for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model, mel, sd = bench.build_model('cuda:0', 'bf16')
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
with torch.no_grad():
    for _ in range(steps):
        out = model.run_on_batch({tools.KEY_AUDIO: audio})
torch.cuda.synchronize()
print('done', out[tools.KEY_ONSETS].shape)
