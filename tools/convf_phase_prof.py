#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused convolution stack (convf.hip), debug build only:
    touch amt_tools_amd/csrc/convf.hip; AMTX_EXTRA_FLAGS=-DAMTX_CONVF_TIMING python -m amt_tools_amd.build
    python tools/convf_phase_prof.py [clips]
Prints the average cycles per step wave 0 (a producer) and wave 4 (a consumer) of a block spend in each phase."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools, _lib
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model, mel, sd = bench.build_model('cuda:0', 'bf16')
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(2):
        model.run_on_batch({tools.KEY_AUDIO: audio})
    torch.cuda.synchronize()
    assert L.amtxdbg_convf_prof(buf, 1) == 0
    model.run_on_batch({tools.KEY_AUDIO: audio})
    torch.cuda.synchronize()
    assert L.amtxdbg_convf_prof(buf, 1) == 0
for k, title, names in ((0, 'producer (wave 0)', ['layer2', 'feature staging + layer1 unit (a step ahead)', 'wait at the barrier']),
                        (8, 'consumer (wave 4)', ['layer1 unit(s) (a step ahead)', 'layer3', 'wait at the barrier'])):
    steps = buf[k + 3]
    tot = sum(buf[k + i] for i in range(3))
    print(f'{title}: {steps} steps, cycles per step:')
    for i, n in enumerate(names):
        print(f'   {n:<46} {buf[k + i] / steps:9.0f}  ({100.0 * buf[k + i] / tot:4.1f} %)')
    print(f'   total                                          {tot / steps:9.0f}')
