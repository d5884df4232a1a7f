#!/usr/bin/env python3
"""RCCL under test on a one-GPU box: the config-4 training step (amt_tools/train.py:126-141 around
amt_tools_amd.dp.DataParallelOptimizer) with a ONE-RANK `nccl` process group and the flat all-reduce forced on, beside the
persistent HIP BiLSTM autograd kernels, for `--steps` steps -- and the same steps without any collective, from the same seed.
A sum over one rank returns the gradients' own bits, so both runs must leave bit-identical weights.

    python tools/rccl_single_rank.py [--steps 300] [--of2] [--clips 8] [--frames 625]

The default training path runs the pitch head on a side stream beside the recurrent heads (models.py); a third run keeps everything on
ONE stream -- same kernels, same order per stream, so again the same bits (a cross-stream race would show as a difference).

Prints ONE JSON line: {"steps", "model", "identical", "max_abs_diff", "collectives", "ms_per_step_plain", "ms_per_step_rccl",
"deterministic_plain", "ms_per_step_one_stream", "two_streams_identical_to_one_stream"}.  Meant to be started as a FRESH child process under a wall-clock limit (tests/test_gpu_rccl.py): a hang
is then a killed child and a failed test, never a re-exec of a process that has touched the GPU.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402


def run(steps, of2, clips, frames, collective, overlap=True, device='cuda:0'):
    from amt_tools_amd import tools
    from amt_tools_amd.dp import DataParallelOptimizer
    from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
    from amt_tools_amd.synth import synth_labels
    torch.manual_seed(0)
    torch.cuda.manual_seed_all(0)
    if of2:
        model = OnsetsFrames2(229, tools.PianoProfile(), 1, device=device)
    else:
        model = OnsetsFrames(229, tools.PianoProfile(), 1, 2, device=device)
    model.change_device()
    model.train()
    model.__dict__['overlap_heads'] = bool(overlap)       # pitch head on a side stream beside the recurrent heads (models.py)
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=6e-4, buffers=model.buffers(), force_collective=collective)
    rng = np.random.default_rng(7)
    feats = torch.from_numpy(rng.random((clips, 1, 229, frames), dtype=np.float32)).to(device)
    lab = [synth_labels(i, num_frames=frames) for i in range(clips)]
    batch = {tools.KEY_FEATS: feats,
             tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])).to(device),
             tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab])).to(device)}
    if of2:
        batch[tools.KEY_OFFSETS] = torch.from_numpy(np.stack([l[1][:, ::-1].copy() for l in lab])).to(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):                       # amt_tools/train.py:122-141
        opt.zero_grad()
        loss = model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return state, ms, opt.collectives_run, float(loss)


def diff(a, b):
    worst, same = 0.0, True
    for k in a:
        if not torch.equal(a[k], b[k]):
            same = False
            if a[k].dtype.is_floating_point:
                worst = max(worst, float((a[k].float() - b[k].float()).abs().max()))
    return same, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--of2', action='store_true')
    ap.add_argument('--clips', type=int, default=8)
    ap.add_argument('--frames', type=int, default=625)
    ap.add_argument('--backend', default='nccl')
    args = ap.parse_args()
    assert torch.cuda.is_available()
    import torch.distributed as dist
    torch.cuda.set_device(0)
    plain, ms_plain, n0, loss_plain = run(args.steps, args.of2, args.clips, args.frames, collective=False)
    one, ms_one, _, _ = run(args.steps, args.of2, args.clips, args.frames, collective=False, overlap=False)
    same_streams, worst_streams = diff(plain, one)
    again, _, _, _ = run(min(args.steps, 20), args.of2, args.clips, args.frames, collective=False)
    first, _, _, _ = run(min(args.steps, 20), args.of2, args.clips, args.frames, collective=False)
    deterministic, _ = diff(again, first)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    else:
        dist.init_process_group(args.backend, rank=0, world_size=1)
    rccl, ms_rccl, n1, loss_rccl = run(args.steps, args.of2, args.clips, args.frames, collective=True)
    same, worst = diff(plain, rccl)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps({'steps': args.steps, 'model': 'OnsetsFrames2(mc=3)' if args.of2 else 'OnsetsFrames(mc=2)', 'backend': args.backend,
                      'identical': same, 'max_abs_diff': worst, 'collectives': n1, 'collectives_plain': n0,
                      'ms_per_step_plain': ms_plain, 'ms_per_step_rccl': ms_rccl, 'deterministic_plain': deterministic,
                      'ms_per_step_one_stream': ms_one, 'two_streams_identical_to_one_stream': same_streams, 'max_abs_diff_streams': worst_streams,
                      'loss_plain': loss_plain, 'loss_rccl': loss_rccl}), flush=True)
    return 0 if (same and same_streams and n1 == args.steps and n0 == 0) else 1


if __name__ == '__main__':
    sys.exit(main())
