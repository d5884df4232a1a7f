#!/usr/bin/env python3
"""BASELINE config 3 throughput: audio (resident in HBM) -> HIP HCQT (6 harmonics x 72 bins) as model.frontend ->
OnsetsFrames(dim_in=72, in_channels=6).  Usage: python tools/bench_hcqt.py [clips=128] [precision=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from amt_tools_amd import tools, _lib
from amt_tools_amd.features import HCQT
from amt_tools_amd.models import OnsetsFrames
from amt_tools_amd.synth import synth_clip, synth_state_dict, CLIP_FRAMES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = 'cuda:0'
mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, device=dev)
PREC = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device=dev, precision=PREC)
sd = synth_state_dict(0, dim_in=72, in_channels=6, model_complexity=2)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model.frontend = torch.nn.Sequential(mod.frontend())
model.change_device(); model.eval()
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).to(dev).repeat((B + 3) // 4, 1)[:B].contiguous()
def step():
    with torch.no_grad():
        return model.run_on_batch({tools.KEY_AUDIO: audio})
for _ in range(2): out = step()
torch.cuda.synchronize()
# front-end alone
t0 = time.perf_counter()
feats16 = model._get_engine(torch.device(dev)).takes_feats16() and os.environ.get('AMTX_CQT_FEATS16', '1') != '0'
with torch.no_grad(): f = mod.process_batch16(audio) if feats16 else model.frontend(audio[:, None, :])    # untimed: first call of this form
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    with torch.no_grad(): f = mod.process_batch16(audio) if feats16 else model.frontend(audio[:, None, :])
torch.cuda.synchronize()
fe = (time.perf_counter() - t0) / 3
L = _lib.lib(); eng = model._get_engine(torch.device(dev))
_lib.check(L.amtx_of_profile_enable(eng.handle, 1))
t0 = time.perf_counter()
for _ in range(5): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
ms = (C.c_double * L.amtx_of_num_stages())(); n = C.c_int(0)
_lib.check(L.amtx_of_profile_read(eng.handle, ms, C.byref(n)))
T = out[tools.KEY_ONSETS].shape[-1]
print(f'{B} clips x {T} frames: {dt * 1e3:.2f} ms/step = {B * T / dt / 1e6:.2f} M frames/s; HCQT front-end alone {fe * 1e3:.2f} ms; engine stages (ms): ' +
      ', '.join(f'{L.amtx_of_stage_name(i).decode()} {ms[i] / max(1, n.value):.2f}' for i in range(L.amtx_of_num_stages()) if ms[i] > 0))
