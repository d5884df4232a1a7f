#!/usr/bin/env python3
"""Per-section cycle shares of the tuned spectrogram kernel (debug build only):
    touch amt_tools_amd/csrc/spec.hip; AMTX_EXTRA_FLAGS=-DAMTX_SPEC_TIMING python -m amt_tools_amd.build
    python tools/spec_phase_prof.py [clips]
Every tick is an s_memtime (a scalar-memory round trip): read the SHARES, not the absolute cycles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import _lib
from amt_tools_amd.features import MelSpec
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ.setdefault('AMTX_SPEC_NO_RING', '1')   # the counters live in the general kernel
mel = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)
audio = torch.from_numpy(np.stack([synth_clip(i) for i in range(4)])).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
for _ in range(2):
    mel.power_batch(audio)
torch.cuda.synchronize()
assert L.amtxdbg_spec_prof(buf, 1) == 0
mel.power_batch(audio)
torch.cuda.synchronize()
assert L.amtxdbg_spec_prof(buf, 1) == 0
names = ['window + next-frame load issue', '-', '-', 'FFT (two DFT-16 passes, two exchanges) + radix-4 tail + untangling', 'mel gather + stores']
tot = sum(buf[i] for i in range(5))
for i, n in enumerate(names):
    print(f'   {n:<40} {buf[i] / max(1, buf[5]):8.0f} cycles / frame  ({100.0 * buf[i] / tot:4.1f} %)')
print(f'   total {tot / max(1, buf[5]):.0f} cycles per frame of wave 0, {buf[5]} frames')
