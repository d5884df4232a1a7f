#!/usr/bin/env python3
"""Per-phase cycle counters of the CQT basis-product kernel (cqt.hip, cqt_basis_kernel), debug build only:
    tools/build_dbg.sh cqtbasis cqt.hip -DAMTX_CQT_TIMING && AMTX_LIB_PATH=tools/_dbg/libamtx_cqtbasis.so python tools/cqt_basis_prof.py [clips=512]
Prints the cycles per tile and phase of wave 0 over the basis products of one HCQT (BASELINE config 3) front-end call."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import _lib
from amt_tools_amd.features import HCQT
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = _lib.lib()
prof = L.amtxdbg_bas_prof
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
names = ['tile setup', 'staging (loads, split, LDS stores)', 'barrier', 'matrix loop + magnitude stores', 'maxima + barrier + atomics']
n = max(1, v[7])
print(f'basis products of one HCQT call, {B} clips: {v[7]} tiles (wave 0 of every block), {sum(v[:5]) / n:.0f} cycles per tile')
for i in range(5):
    print(f'  {names[i]:36s} {100.0 * v[i] / sum(v[:5]):5.1f} %  {v[i] / n:8.0f} cycles per tile')
