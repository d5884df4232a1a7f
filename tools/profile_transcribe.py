#!/usr/bin/env python3
"""Host-side profile of the batched transcription driver (BASELINE config 5 on one GPU): cProfile over one warm run of
run_offline_batched, sorted by own time.  Usage: python tools/profile_transcribe.py [num_clips=2048] [batch=512]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import bench            # noqa: E402
from amt_tools_amd.inference import run_offline_batched     # noqa: E402
from amt_tools_amd.synth import synth_clip, CLIP_FRAMES     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, mel, sd = bench.build_model('cuda:0', 'bf16')
base = np.stack([synth_clip(i) for i in range(8)])
host = torch.from_numpy(np.tile(base, ((N + 7) // 8, 1))[:N]).pin_memory()
times = np.arange(CLIP_FRAMES) * 512 / 22050.0
model.frontend = torch.nn.Sequential(mel.frontend())


def run():
    res = run_offline_batched(host, model, times=times, batch_size=B, decode_notes=True, keep=())
    torch.cuda.synchronize()
    return res


run()
t0 = time.perf_counter()
run()
print(f'{N} clips, batches of {B}: {(time.perf_counter() - t0) * 1e3:.1f} ms unprofiled')
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
