#!/usr/bin/env python3
"""Training-step timing (BASELINE config 4): OnsetsFrames(mc=2) fwd + bwd + Adam on 8 clips x 625 frames per GPU,
clip-level data parallelism through amt_tools_amd.dp.DataParallelOptimizer (one flat gradient all-reduce per step;
backend nccl = RCCL).  Features come from the HIP mel front-end in `model.frontend`; the model's forward/backward in
training mode is ATen autograd for the convolutions / Linear layers, the HIP BiLSTM autograd function (amt_tools_amd/autograd.py:
one persistent kernel forward, one backward) for the three recurrences and the HIP BatchNorm(batch statistics)+ReLU+MaxPool passes.

    python tools/bench_train.py [--steps K] [--warmup W] [--clips 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from amt_tools_amd import tools
from amt_tools_amd.dp import DataParallelOptimizer, broadcast_parameters, init_distributed
from amt_tools_amd.features import MelSpec
from amt_tools_amd.models import OnsetsFrames
from amt_tools_amd.synth import synth_clip, synth_labels

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--warmup', type=int, default=3)
ap.add_argument('--clips', type=int, default=8)
ap.add_argument('--of2', action='store_true', help='OnsetsFrames2 as shipped (model_complexity 3, offset head, detach_heads) instead of OnsetsFrames(mc=2)')
args = ap.parse_args()

rank, world, device = init_distributed()
torch.manual_seed(0)
if args.of2:
    from amt_tools_amd.models import OnsetsFrames2
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, device=str(device))
else:
    model = OnsetsFrames(229, tools.PianoProfile(), 1, 2, device=str(device))
model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, device=str(device)).frontend())
model.change_device()
broadcast_parameters(model)
model.train()
opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=6e-4)
B = args.clips
audio = torch.from_numpy(np.stack([synth_clip(rank * B + i) for i in range(B)])).to(device)
lab = [synth_labels(rank * B + i) for i in range(B)]
batch = {tools.KEY_AUDIO: audio,
         tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])).to(device),
         tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab])).to(device)}
if args.of2:
    batch[tools.KEY_OFFSETS] = torch.from_numpy(np.stack([l[1][:, ::-1].copy() for l in lab])).to(device)   # any sparse binary map


def step():
    opt.zero_grad()
    loss = model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
    loss.backward()
    opt.step()
    return loss


for _ in range(args.warmup):
    loss = step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
if rank == 0:
    print(json.dumps({'metric': 'train step time (%s fwd+bwd+Adam, 8 clips x 625 frames per GPU)' % ('OnsetsFrames2 mc=3' if args.of2 else 'OnsetsFrames'), 'value': dt / args.steps * 1e3,
                      'unit': 'ms/step', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': False,
                      'scaling': 'weak', 'frames_per_s': world * B * 625 * args.steps / dt, 'loss': float(loss),
                      'backward': 'ATen autograd (conv / linear) + HIP BiLSTM and BatchNorm+ReLU+MaxPool forward/backward kernels'}))
if world > 1:
    dist.destroy_process_group()
