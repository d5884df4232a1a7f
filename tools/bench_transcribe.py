#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, host to host: synthetic clips in pinned host memory -> H2D -> log-mel + Onsets & Frames
(bf16) -> device note decoding -> note lists (and optionally piano rolls) back on the host.  This is the PCIe-inclusive rate
DESIGN.md quotes next to bench.py's HBM-resident `value`; the next batch's upload runs on a copy stream under the current
batch's kernels.
Usage: python tools/bench_transcribe.py [num_clips=2048] [batch=512] [--rolls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip, CLIP_FRAMES
from amt_tools_amd.transcribe import decode_notes_batch_async

args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if args else 2048
B = int(args[1]) if len(args) > 1 else 512
rolls = '--rolls' in sys.argv
model, mel, sd = bench.build_model('cuda:0', 'bf16')
base = np.stack([synth_clip(i) for i in range(8)])
host = torch.from_numpy(np.tile(base, ((N + 7) // 8, 1))[:N]).pin_memory()
times = np.arange(CLIP_FRAMES) * 512 / 22050.0
copy_stream = torch.cuda.Stream()


def upload(i):
    with torch.cuda.stream(copy_stream):
        d = host[i:i + B].to('cuda:0', non_blocking=True)
        ev = torch.cuda.Event(); ev.record(copy_stream)
    return d, ev


def run():
    notes_total, pending = 0, None
    nxt = upload(0)

    def finish(p):
        handle, rolls_dev = p
        n = sum(len(x) for x in handle.result())
        if rolls_dev is not None:
            _ = [r.cpu() for r in rolls_dev]
        return n

    with torch.no_grad():
        for i in range(0, N, B):
            audio, ev = nxt
            torch.cuda.current_stream().wait_event(ev)
            if i + B < N:
                nxt = upload(i + B)
            preds = model.run_on_batch({tools.KEY_AUDIO: audio})
            handle = decode_notes_batch_async(preds[tools.KEY_ONSETS], preds[tools.KEY_MULTIPITCH], times, 21)
            # this batch's kernels are enqueued; the host now assembles the PREVIOUS batch's notes while the GPU works
            if pending is not None:
                notes_total += finish(pending)
            pending = (handle, (preds[tools.KEY_ONSETS], preds[tools.KEY_MULTIPITCH]) if rolls else None)
        notes_total += finish(pending)
    torch.cuda.synchronize()
    return notes_total


run()
t0 = time.perf_counter()
total = run()
dt = time.perf_counter() - t0
print(f'{N} clips x {CLIP_FRAMES} frames, batches of {B}, host audio -> host notes{" + piano rolls" if rolls else ""}: {dt * 1e3:.1f} ms, '
      f'{N / dt:.0f} clips/s, {N * CLIP_FRAMES / dt / 1e6:.2f} M frames/s ({total} notes); H2D {host.numel() * 4 / 1e9:.2f} GB')
