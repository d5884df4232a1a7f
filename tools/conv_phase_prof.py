#!/usr/bin/env python3
"""Per-phase cycle breakdown of the persistent conv kernels (debug build only):
    AMTX_EXTRA_FLAGS=-DAMTX_CONV_TIMING python -m amt_tools_amd.build   (touch conv.hip first)
    python tools/conv_phase_prof.py [clips]
Prints, per kernel, the average cycles wave 0 of a block spends per tile in each phase."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools, _lib
from amt_tools_amd.synth import synth_clip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
model, mel, sd = bench.build_model("cuda:0", PREC)
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 32)()
with torch.no_grad():
    for _ in range(2):
        model.run_on_batch({tools.KEY_AUDIO: audio})
    torch.cuda.synchronize()
    assert L.amtxdbg_conv_prof(buf, 1) == 0
    model.run_on_batch({tools.KEY_AUDIO: audio})
    torch.cuda.synchronize()
    assert L.amtxdbg_conv_prof(buf, 1) == 0
names = ['stage/feature store', 'barrier', 'first-conv phase', 'barrier', 'conv phase', 'trailing barrier']
for k, title in ((0, 'fused conv1+conv2'), (8, 'conv3')):
    tiles, blocks = buf[k + 6], buf[k + 7]
    if not tiles:
        continue
    print(f'{title}: {blocks} blocks, {tiles} tiles, cycles per tile (wave 0):')
    tot = sum(buf[k + i] for i in range(6))
    for i, n in enumerate(names):
        print(f'   {n:<22} {buf[k + i] / tiles:9.0f}  ({100.0 * buf[k + i] / tot:4.1f} %)')
    print(f'   total                  {tot / tiles:9.0f}')

it = buf[22]
if it:
    secs = ['pack gathered values', 'gather issue (next)', 'acc init + MFMA issue', 'epilogue group 0', 'epilogue group 1']
    print(f'first-conv loop, cycles per iteration (wave 0, {it} iterations, {it / max(1, buf[6]):.2f} per tile):')
    for i, n in enumerate(secs):
        print(f'   {n:<24} {buf[16 + i] / it:8.0f}')
