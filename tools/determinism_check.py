#!/usr/bin/env python3
"""Run-to-run determinism at bench sizes: the same batch through run_on_batch N times, every output compared bit for bit with the first --
config 3 (HCQT 6 x 72, bf16 and x3, 512 clips) and the headline (mel, bf16, 1024 clips).  A race in a kernel that the small parity tests do
not provoke (a missing barrier, a DMA landing late) shows here as a difference.  Usage: python tools/determinism_check.py [runs=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import tools
from amt_tools_amd.features import HCQT, MelSpec
from amt_tools_amd.models import OnsetsFrames
from amt_tools_amd.synth import synth_clip, synth_state_dict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = 'cuda:0'
base = np.stack([synth_clip(i) for i in range(8)])
ok = True
for name, mod, F, C, prec, B in (('config 3 bf16', HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, device=dev), 72, 6, 'bf16', 512),
                                 ('config 3 x3', HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, device=dev), 72, 6, 'x3', 512),
                                 ('headline bf16', MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048, device=dev), 229, 1, 'bf16', 1024)):
    model = OnsetsFrames(F, tools.PianoProfile(), C, 2, device=dev, precision=prec)
    sd = synth_state_dict(0, dim_in=F, in_channels=C, model_complexity=2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.frontend = torch.nn.Sequential(mod.frontend())
    model.change_device(); model.eval()
    audio = torch.from_numpy(base).to(dev).repeat((B + 7) // 8, 1)[:B].contiguous()
    T = None
    lab = None
    first = None
    bad = 0
    for r in range(N):
        with torch.no_grad():
            if lab is None:
                out = model.run_on_batch({tools.KEY_AUDIO: audio})
                T = out[tools.KEY_ONSETS].shape[-1]
                lab = {tools.KEY_MULTIPITCH: torch.zeros(B, 88, T, device=dev), tools.KEY_ONSETS: torch.zeros(B, 88, T, device=dev)}
            out = model.run_on_batch(dict({tools.KEY_AUDIO: audio}, **lab))          # labelled: the raw logits come back
        cur = [out[k].clone() for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH)]
        if first is None:
            first = cur
        elif not all(torch.equal(a, b) for a, b in zip(first, cur)):
            bad += 1
    print(f'{name}: {N} runs of {B} clips x {T} frames, {bad} differ from the first')
    ok = ok and bad == 0
    del model, audio, first, cur, out, lab
    torch.cuda.empty_cache()
sys.exit(0 if ok else 1)
