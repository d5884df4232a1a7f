#!/usr/bin/env python3
"""Per-phase cycle counters of the general-channel conv kernel (convg.hip), debug build only:
    AMTX_EXTRA_FLAGS=-DAMTX_CONV_TIMING python -m amt_tools_amd.build && python tools/convg_phase_prof.py [clips=128] [hcqt]
(or AMTX_LIB_PATH=<a library whose convg.o was compiled with -DAMTX_CONV_TIMING>).
Prints, for conv2 (fused first conv) and conv3 of OnsetsFrames2(mc=3) -- or, with `hcqt`, for conv2 of the BASELINE config-3 model
(OnsetsFrames mc 2, 6 input channels x 72 bins; its conv3 runs on conv.hip) -- the share of wave 0's cycles per phase."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amt_tools_amd import tools, _lib
from amt_tools_amd.models import OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = _lib.lib()
prof = L.amtxdbg_convg_prof
prof.restype = C.c_int; prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
if 'hcqt' in sys.argv[2:]:
    from amt_tools_amd.models import OnsetsFrames
    model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device='cuda:0', precision='bf16')
    sd = synth_state_dict(0, dim_in=72, in_channels=6, model_complexity=2)
    feats = torch.rand(B, 6, 72, 625, device='cuda:0')
else:
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, 3, device='cuda:0', precision='bf16')
    sd = synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
    feats = torch.rand(B, 1, 229, 625, device='cuda:0')
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model.change_device(); model.eval()
with torch.no_grad():
    model.engine_logits(feats)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
prof(buf, 1)
with torch.no_grad():
    model.engine_logits(feats)
torch.cuda.synchronize()
prof(buf, 1)
v = list(buf)
names = ['tile store + barrier', 'first conv', 'weights + barrier', 'MFMA loop', 'epilogue stores', 'trailing barrier', 'chunk-tiles', 'tile setup']
tot = sum(v[:6]) + v[7]
print(f'conv2 + conv3 of one forward, {B} clips: {v[6]} chunk-tiles, {tot / max(1, v[6]):.0f} cycles per chunk-tile (wave 0)')
for i in (7, 0, 1, 2, 3, 4, 5):
    print(f'  {names[i]:24s} {100.0 * v[i] / tot:5.1f} %   {v[i] / max(1, v[6]):8.0f} cycles per chunk-tile')
