#!/usr/bin/env python3
"""Throughput of the reference's OnsetsFrames2 experiment shape (scripts of_2.py:87-110): audio resident in HBM ->
HIP log-mel (229 bins, HTK spacing) as model.frontend -> OnsetsFrames2(model_complexity=3: 48/48/96-channel
convolutions, fc 768, BiLSTM hidden 256, offset head), bf16.  93.6 MFLOP per clip-frame.
Usage: python tools/bench_of2.py [clips=256] [model_complexity=3] [precision=bf16|f16|x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from amt_tools_amd import tools, _lib
from amt_tools_amd.features import MelSpec
from amt_tools_amd.models import OnsetsFrames2
from amt_tools_amd.synth import synth_clip, synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MC = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PREC = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
dev = 'cuda:0'
mel = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048, htk=True, device=dev)
model = OnsetsFrames2(229, tools.PianoProfile(), 1, MC, device=dev, precision=PREC)
sd = synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=MC, offsets=True)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model.frontend = torch.nn.Sequential(mel.frontend())
model.change_device(); model.eval()
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).to(dev).repeat((B + 3) // 4, 1)[:B].contiguous()
def step():
    with torch.no_grad():
        return model.run_on_batch({tools.KEY_AUDIO: audio})
for _ in range(2): out = step()
torch.cuda.synchronize()
L = _lib.lib(); eng = model._get_engine(torch.device(dev))
_lib.check(L.amtx_of_profile_enable(eng.handle, 1))
N = 5
t0 = time.perf_counter()
for _ in range(N): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
ms = (C.c_double * L.amtx_of_num_stages())(); n = C.c_int(0)
_lib.check(L.amtx_of_profile_read(eng.handle, ms, C.byref(n)))
T = out[tools.KEY_ONSETS].shape[-1]
def _macs(mc, F=229, n_out=88):
    # multiply-adds per clip-frame of the reference's layers (onsetsframes.py:375-427, 498-507, common.py:539): three acoustic models,
    # two recurrent heads, the refinement stage
    nf1, nf3, am, lm = 16 * mc, 32 * mc, 256 * mc, 256 * (mc - 1)
    acoustic = 9 * nf1 * F + 9 * nf1 * nf1 * F + 9 * nf1 * nf3 * (F // 2) + nf3 * (F // 4) * am
    lstm = lambda d_in: d_in * 4 * lm + (lm // 2) * 4 * lm + lm * n_out
    return 3 * acoustic + 2 * lstm(am) + am * n_out + lstm(3 * n_out)
mac = {2: 13347648 + 6145000, 3: 46.8e6}.get(MC) or _macs(MC)     # per clip-frame incl. the offset head (mc 3: SURVEY 8d 93.6 MFLOP)
fps = B * T / dt
print(f'OnsetsFrames2(mc={MC}{"" if PREC == "bf16" else ", " + PREC}) {B} clips x {T} frames: {dt * 1e3:.2f} ms/step = {fps / 1e6:.2f} M frames/s'
      + (f' = {2 * mac * fps / 1e12:.0f} TFLOP/s ({2 * mac * fps / 2.5e15:.1%} of the 2.5 PF dense bf16 peak)' if mac else '')
      + '; engine stages (ms): ' + ', '.join(f'{L.amtx_of_stage_name(i).decode()} {ms[i] / max(1, n.value):.2f}'
                                               for i in range(L.amtx_of_num_stages()) if ms[i] > 0))
