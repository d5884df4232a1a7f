#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, host to host: synthetic clips in pinned host memory -> H2D -> log-mel + Onsets & Frames
(bf16) -> device note decoding -> note lists (and optionally piano rolls) back on the host.  This is the PCIe-inclusive rate
DESIGN.md quotes next to bench.py's HBM-resident `value`; the next batch's upload runs on a copy stream under the current
batch's kernels.
A long run amortises the pipeline's fill and drain (one upload and one batch of compute + host work that nothing overlaps); the
steady state is bound by the upload: 2 KB per frame over a measured 57 GB/s link = 27.8 M frames/s.  Two warm-up runs, then the
median and the best of three timed runs (single runs on a shared host vary by 2x).
Usage: python tools/bench_transcribe.py [num_clips=8192] [batch=256] [--rolls] [--pcm16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip, CLIP_FRAMES

args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if args else 8192
B = int(args[1]) if len(args) > 1 else 256
rolls = '--rolls' in sys.argv
pcm16 = '--pcm16' in sys.argv      # the clips as 16-bit PCM (what the audio files hold): half the upload
model, mel, sd = bench.build_model('cuda:0', 'bf16')
base = np.stack([synth_clip(i) for i in range(8)])
if pcm16:
    base = np.clip(np.round(base / np.abs(base).max() * 32767.0), -32768, 32767).astype(np.int16)
host = torch.from_numpy(np.tile(base, ((N + 7) // 8, 1))[:N]).pin_memory()
times = np.arange(CLIP_FRAMES) * 512 / 22050.0
model.frontend = torch.nn.Sequential(mel.frontend())
from amt_tools_amd.inference import run_offline_batched
keep = (tools.KEY_ONSETS, tools.KEY_MULTIPITCH) if rolls else ()


def run():
    res = run_offline_batched(host, model, times=times, batch_size=B, decode_notes=True, keep=keep)
    torch.cuda.synchronize()
    return sum(len(r[tools.KEY_NOTES]) for r in res.values())


run()
run()
dts = []
for _ in range(3):
    t0 = time.perf_counter()
    total = run()
    dts.append(time.perf_counter() - t0)
dt, best = sorted(dts)[1], min(dts)
print(f'{N} clips x {CLIP_FRAMES} frames, batches of {B}, host audio -> host notes{" + piano rolls" if rolls else ""}: {dt * 1e3:.1f} ms, '
      f'{N / dt:.0f} clips/s, {N * CLIP_FRAMES / dt / 1e6:.2f} M frames/s median of 3 (best {N * CLIP_FRAMES / best / 1e6:.2f} M; {total} notes); '
      f'H2D {host.numel() * host.element_size() / 1e9:.2f} GB{" (int16 PCM)" if pcm16 else ""}')
