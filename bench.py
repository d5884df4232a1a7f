#!/usr/bin/env python3
"""
Headline benchmark: audio frames/s of OnsetsFrames(mc=2) + MelSpec-229 inference (BASELINE.json configs[1])
on synthetic 22.05 kHz clips of 319 999 samples (625 frames), bf16 MFMA arithmetic.

One "step" = one pass of the hot path over one batch of clips already resident in HBM:
    audio (B,N) -> HIP mel front-end -> HIP Onsets&Frames engine -> piano rolls (B,88,T) x2
through the product API (`model.run_on_batch({'audio': ...})` with the front-end in `model.frontend`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--clips B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, clips sharded across ranks, no data-path collective (inference shares nothing);
the only collectives are the timing barrier and the MAX over ranks of the elapsed time.  Weak scaling.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline     -- dominant kernel of the timed region, timed live with HIP events on the launch stream
  cpu_baseline -- the oracle (numpy front-end + torch-CPU fp32 model restatement) on a bounded sample
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

SR, HOP, N_MELS, N_FFT = 22050, 512, 229, 2048
CLIP_SAMPLES, CLIP_FRAMES = 319999, 625
PEAK_MFMA_BF16_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

# algorithmic work per clip-frame; flops for MFMA stages, bytes for streaming stages
STAGE_FLOPS = {
    'conv1': 2 * 2 * 229 * 32 * 9,
    'conv2_pool': 2 * 2 * 229 * 32 * 32 * 9,
    'conv3_pool': 2 * 2 * 114 * 32 * 64 * 9,
    'fc1_gemm': 2 * 2 * 3648 * 512,
    'rec_xproj_gemm': 2 * 512 * 1024,
    'rec_bilstm': 2 * 2 * 512 * 128,
    'rec_head_gemm': 2 * 256 * 88,
    'pitch_head_gemm': 2 * 512 * 88,
    'adj_xproj_gemm': 2 * 176 * 1024,
    'adj_bilstm': 2 * 2 * 512 * 128,
    'adj_head_gemm': 2 * 256 * 88,
}
# compulsory HBM bytes per clip-frame of the MFMA stages (bf16 activations; weights are L2-resident)
ALGO_BYTES = {
    'conv2_pool': 2 * (229 * 4 + 114 * 32 * 2),         # both heads: read the log-mel row, write the pooled 114 x 32 map
    'conv3_pool': 2 * (114 * 32 * 2 + 57 * 64 * 2),
    'fc1_gemm': 2 * (3648 * 2 + 512 * 2),
}
STAGE_BYTES = {
    'spec_power': HOP * 4 + N_MELS * 4,            # read hop samples, write the mel-power row
    'spec_scale': N_MELS * 4 * 2,                  # read mel power, write scaled features
    'pianoroll': 2 * 88 * 4 * 2,                   # read 2x88 logits, write 2x88 outputs
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--clips', type=int, default=1024, help='clips per GPU per step (sweep on MI355X: 256: 26.7, 512: 29.1, 1024: 31.1, 2048: 31.1 M frames/s)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'x3'])
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='wall-time budget of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only for the "
                                                      "single-GPU smoke test of the multi-process path)")
    ap.add_argument('--share-device', action='store_true', help='test only: every rank uses cuda:0')
    return ap.parse_args()


def build_model(device, precision):
    from amt_tools_amd import tools
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd.synth import synth_state_dict
    model = OnsetsFrames(N_MELS, tools.PianoProfile(), 1, 2, device=device, precision=precision)
    sd = synth_state_dict(0, dim_in=N_MELS, in_channels=1, model_complexity=2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    mel = MelSpec(sample_rate=SR, hop_length=HOP, n_mels=N_MELS, n_fft=N_FFT, device=device)
    model.frontend = torch.nn.Sequential(mel.frontend())
    model.change_device()
    model.eval()
    return model, mel, sd


def cpu_baseline(budget_s, sd):
    """Oracle on the host cores: numpy front-end restatement + torch-CPU fp32 model restatement, one clip
    per call (the reference's run_offline pattern, amt_tools/inference.py:38-41).  Bounded by wall time:
    clips are processed until `budget_s` is used up (at least one, at most 64).  The thread count is capped
    at 16: the per-step LSTM matmuls are tiny and more threads only add synchronisation cost."""
    from amt_tools_amd.synth import synth_clip
    from oracle import frontend_np as fe, model_ref
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    t0 = time.perf_counter()
    t_fe, n = 0.0, 0
    while n < 64:
        y = synth_clip(n)
        a = time.perf_counter()
        feats = fe.melspec_process_audio(y, SR, HOP, N_MELS, N_FFT, dtype=np.float32).astype(np.float32)
        t_fe += time.perf_counter() - a
        with torch.no_grad():
            model_ref.run_on_batch(torch.from_numpy(feats[None]), sdt)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {'value': n * CLIP_FRAMES / dt, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} synthetic clips x {CLIP_FRAMES} frames in {dt:.1f} s, one clip per call, fp32 '
                      f'(front-end share {t_fe / dt:.2f}; includes clip synthesis)'}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)

    from amt_tools_amd import _lib, tools
    from amt_tools_amd.synth import synth_clip

    model, mel, sd = build_model(device, args.precision)
    B = args.clips
    base = np.stack([synth_clip(rank * 8 + i) for i in range(8)])
    audio = torch.from_numpy(base).to(device).repeat((B + 7) // 8, 1)[:B].contiguous()
    batch = {tools.KEY_AUDIO: audio}

    def step():
        with torch.no_grad():
            return model.run_on_batch(batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        out = step()
    assert out[tools.KEY_ONSETS].shape == (B, 88, CLIP_FRAMES)

    # per-kernel HIP events on the launch stream, live in the timed region
    L = _lib.lib()
    eng = model._get_engine(torch.device(device))
    _lib.check(L.amtx_of_profile_enable(eng.handle, 1))
    mel._prof_events = []

    # a generation-2 garbage collection of the interpreter (tens of ms with torch + numpy loaded) that lands in the first
    # steps of the timed loop stalls the enqueueing thread long enough for the GPU queue to run dry: seen as 15 instead of
    # 11.4 ms/step with identical per-kernel times.  Collect now, keep the collector off while timing.
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage_ms = (C.c_double * L.amtx_of_num_stages())()
    nfw = C.c_int(0)
    _lib.check(L.amtx_of_profile_read(eng.handle, stage_ms, C.byref(nfw)))
    _lib.check(L.amtx_of_profile_enable(eng.handle, 0))
    per_launch = {L.amtx_of_stage_name(i).decode(): stage_ms[i] / max(1, nfw.value) for i in range(L.amtx_of_num_stages())}
    fe_ms = {}
    for name, e0, e1 in mel._prof_events:
        fe_ms.setdefault(name, []).append(e0.elapsed_time(e1))
    mel._prof_events = None
    for name, v in fe_ms.items():
        per_launch[name] = float(np.mean(v))

    if rank == 0:
        frames_per_launch = B * CLIP_FRAMES
        total_frames = world * B * CLIP_FRAMES * args.steps
        dom = max(per_launch, key=per_launch.get)
        dur_s = per_launch[dom] * 1e-3
        if dom in STAGE_FLOPS:
            ach = STAGE_FLOPS[dom] * frames_per_launch / dur_s / 1e12
            roof = {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_MFMA_BF16_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / PEAK_MFMA_BF16_TFLOPS, 'traffic': None}
        else:
            ach = STAGE_BYTES[dom] * frames_per_launch / dur_s / 1e9
            roof = {'kernel': dom, 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': ach / PEAK_HBM_GBS, 'traffic': None}
        # HBM traffic of the same kernel from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md), scaled to this launch's clips
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as f:
                pmc = json.load(f)
            if dom in pmc:
                roof['traffic'] = pmc[dom]['hbm_bytes_corrected'] * B / pmc[dom]['clips']
                roof['traffic_unit'] = 'bytes per launch'
                roof['algorithmic_bytes'] = ALGO_BYTES.get(dom, 0) * frames_per_launch or None
        except (OSError, ValueError, KeyError):
            pass
        roof['avg_launch_ms'] = per_launch[dom]
        roof['kernel_ms_per_step'] = {k: round(v, 4) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])}
        fps = total_frames / elapsed
        res = {
            'metric': 'audio frames/sec (OnsetsFrames+Mel-229 inference)', 'value': fps, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.precision == 'bf16' else 'bf16x3', 'data': 'synthetic',
            'config': {'workload': 'OnsetsFrames(mc=2)+MelSpec(229 bins, n_fft 2048, hop 512) inference, synthetic 22.05 kHz '
                                   'clips of 319999 samples (625 frames), audio resident in HBM -> piano rolls',
                       'clips_per_gpu_per_step': B, 'frames_per_clip': CLIP_FRAMES, 'parallelism': f'clip-sharded x{world}, no collectives',
                       'whole_path_frac_of_mfma_roof': fps / world * 26.70e6 / 2.5e15,
                       'whole_path_frac_of_compulsory_hbm_roof': fps / world * 2752 / 8.0e12,
                       # a layer-by-layer implementation moves ~125 kB of bf16 activations per frame -> 6.4e7 frames/s
                       # at 8 TB/s (this build moves ~81 kB: PMC, profiles/pmc_traffic.json + the small kernels)
                       'whole_path_frac_of_layerwise_hbm_roof': fps / world * 125.0e3 / 8.0e12},
            'roofline': roof,
        }
        if world == 1 and args.cpu_seconds > 0:
            res['cpu_baseline'] = cpu_baseline(args.cpu_seconds, sd)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
