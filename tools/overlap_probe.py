#!/usr/bin/env python3
"""Does the front-end of batch i + 1 (spec_power: vector-ALU bound) hide beside the model of batch i (matrix-core bound) when the two are
enqueued on two streams?  Prints ms per batch serial | overlapped.  Usage: python tools/overlap_probe.py [clips=1024] [steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = 'cuda:0'
model, mel, sd = bench.build_model(dev, 'bf16')
base = np.stack([synth_clip(i) for i in range(8)])
audio = torch.from_numpy(base).to(dev).repeat((B + 7) // 8, 1)[:B].contiguous()
batch = {tools.KEY_AUDIO: audio}


def serial(n):
    with torch.no_grad():
        for _ in range(n):
            out = model.run_on_batch(dict(batch))
    return out


def overlapped(n):
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    keep = []
    with torch.no_grad():
        with torch.cuda.stream(sA):
            nxt = model.pre_proc(dict(batch)); ev = torch.cuda.Event(); ev.record(sA)
        for i in range(n):
            cur, cur_ev = nxt, ev
            if i + 1 < n:
                with torch.cuda.stream(sA):
                    nxt = model.pre_proc(dict(batch)); ev = torch.cuda.Event(); ev.record(sA)
            with torch.cuda.stream(sB):
                sB.wait_event(cur_ev)
                model.__dict__['_logits_wanted'] = False
                model.__dict__['_in_run_on_batch'] = True
                cur[tools.KEY_OUTPUT] = model(cur[tools.KEY_FEATS])
                out = model.post_proc(cur)
            keep.append(cur)
            if len(keep) > 3:
                keep.pop(0)
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    model.__dict__.pop('_logits_wanted', None); model.__dict__.pop('_in_run_on_batch', None)
    return out


ref = serial(3)
torch.cuda.synchronize()
for name, fn in (('serial', serial), ('overlapped', overlapped), ('serial', serial), ('overlapped', overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn(N)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    same = all(torch.equal(out[k], ref[k]) for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH))
    print(f'{name:10s} {dt * 1e3:.2f} ms/batch  {B * 625 / dt / 1e6:.1f} M frames/s  outputs equal: {same}', flush=True)
