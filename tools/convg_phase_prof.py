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
if 'conv3' in sys.argv[2:]:
    # the 48 -> 96 layer of model_complexity 3 alone (F = 114), through the operator entry point: no fused first conv in the counters
    cin, cout, F = 48, 96, 114
    g = torch.Generator().manual_seed(0)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9.0 * cin) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    n = L.amtx_conv3x3g_packed_elems(cin, cout, 1)
    packed = np.zeros(n, dtype=np.uint16)
    _lib.check(L.amtx_conv3x3g_pack(_lib.ptr(w.numpy()), _lib.ptr(scale.numpy()), cin, cout, 1, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x = torch.rand(B, 625, F, cin, device='cuda:0').to(torch.bfloat16)
    out = torch.empty(B, 625, F // 2, cout, dtype=torch.bfloat16, device='cuda:0')
    shift = torch.zeros(cout, device='cuda:0')

    class _Op(object):
        def engine_logits(self, _):
            _lib.check(L.amtx_conv3x3g_fwd(_lib.ptr(x), 0, _lib.ptr(wp), 1, _lib.ptr(shift), _lib.ptr(out), B, 625, F, cin, cout,
                                           _lib.current_stream(x.device)), 'amtx_conv3x3g_fwd')
    model, feats, sd = _Op(), None, None
elif 'hcqt' in sys.argv[2:]:
    from amt_tools_amd.models import OnsetsFrames
    model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device='cuda:0', precision=('x3' if 'x3' in sys.argv[2:] else 'bf16'))
    sd = synth_state_dict(0, dim_in=72, in_channels=6, model_complexity=2)
    feats = torch.rand(B, 6, 72, 625, device='cuda:0')
else:
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, 3, device='cuda:0', precision='bf16')
    sd = synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
    feats = torch.rand(B, 1, 229, 625, device='cuda:0')
if sd is not None:
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
