#!/usr/bin/env python3
"""Where one ATen training step spends its GPU time (torch.profiler), to decide which backward kernels to hand-write first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from amt_tools_amd import tools
from amt_tools_amd.features import MelSpec
from amt_tools_amd.models import OnsetsFrames
from amt_tools_amd.synth import synth_clip, synth_labels
dev = 'cuda:0'
OF2 = '--of2' in sys.argv           # OnsetsFrames2 as shipped (model_complexity 3, offset head)
if OF2:
    from amt_tools_amd.models import OnsetsFrames2
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, device=dev)
else:
    model = OnsetsFrames(229, tools.PianoProfile(), 1, 2, device=dev)
model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, device=dev).frontend())
model.change_device(); model.train()
opt = torch.optim.Adam(model.parameters(), lr=6e-4)
B = 8
audio = torch.from_numpy(np.stack([synth_clip(i) for i in range(B)])).to(dev)
lab = [synth_labels(i) for i in range(B)]
batch = {tools.KEY_AUDIO: audio, tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])).to(dev),
         tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab])).to(dev)}
if OF2:
    batch[tools.KEY_OFFSETS] = torch.from_numpy(np.stack([l[1][:, ::-1].copy() for l in lab])).to(dev)
def step():
    opt.zero_grad()
    loss = model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = [(e.key, e.self_device_time_total / 3e3, e.count // 3) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(key=lambda r: -r[1])
print('per training step: device-time ms, launches, kernel / op')
for k, ms, n in rows[:28]:
    print(f'{ms:8.3f} {n:5d}  {k[:110]}')
print(f'{sum(r[1] for r in rows):8.3f}        total')
