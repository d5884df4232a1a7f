#!/usr/bin/env python3
"""Per-phase cycle counters of the pyramid's decimator (cqt_dec.hip, cqt_decimate2_kernel), debug build only:
    tools/build_dbg.sh cqttiming cqt_dec.hip -DAMTX_CQT_TIMING && AMTX_LIB_PATH=tools/_dbg/libamtx_cqttiming.so python tools/cqt_phase_prof.py [clips=512]
Prints the cycles per tile and phase of wave 0 (multiplying) and wave 4 (producing) over the seven decimations of one HCQT (BASELINE config 3)
front-end call."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import _lib
from amt_tools_amd.features import HCQT
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = _lib.lib()
prof = L.amtxdbg_dec_prof
prof.restype = C.c_int; prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, device='cuda:0')
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
mod.process_batch(audio)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
prof(buf, 1)
mod.process_batch(audio)
torch.cuda.synchronize()
prof(buf, 1)
v = list(buf)
names = ['multiply: barrier', 'multiply: matrix loop + stores', 'produce: tile setup / pads', 'produce: wait for the DMA', 'produce: split into planes',
         'produce: DMA issue', 'produce: barrier']
n = max(1, v[7])
print(f'decimations of one HCQT call, {B} clips: {v[7]} tiles; cycles per tile: multiplying wave {sum(v[:2]) / n:.0f}, producing wave {sum(v[2:7]) / n:.0f}')
for i in range(7):
    print(f'  {names[i]:32s} {v[i] / n:8.0f} cycles per tile')
