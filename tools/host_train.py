import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import tools
from amt_tools_amd.features import MelSpec
from amt_tools_amd.models import OnsetsFrames
from amt_tools_amd.synth import synth_clip, synth_labels
dev='cuda:0'
model = OnsetsFrames(229, tools.PianoProfile(), 1, 2, device=dev)
model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, device=dev).frontend())
model.change_device(); model.train()
opt = torch.optim.Adam(model.parameters(), lr=6e-4)
B=8
audio = torch.from_numpy(np.stack([synth_clip(i) for i in range(B)])).to(dev)
lab = [synth_labels(i) for i in range(B)]
batch = {tools.KEY_AUDIO: audio, tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])).to(dev), tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab])).to(dev)}
def step():
    opt.zero_grad()
    loss = model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): step()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f'host enqueue time per step {(t1-t0)*100:.2f} ms; with final sync {(t2-t0)*100:.2f} ms per step')
