#!/usr/bin/env python3
"""
Headline benchmark: audio frames/s of OnsetsFrames(mc=2) + MelSpec-229 inference (BASELINE.json configs[1])
on synthetic 22.05 kHz clips of 319 999 samples (625 frames), bf16 MFMA arithmetic.

One "step" = one pass of the hot path over one batch of clips already resident in HBM:
    audio (B,N) -> HIP mel front-end -> HIP Onsets&Frames engine -> piano rolls (B,88,T) x2
through the product API (`model.run_on_batch({'audio': ...})` with the front-end in `model.frontend`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--clips B] [--mode infer|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU.  `python bench.py --gpus N` on its own starts the N ranks itself (fresh child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before this process has touched the GPU) and relays rank 0's JSON line; under
torchrun the ranks already exist.  WORLD_SIZE != --gpus is an error, never a silent 1-GPU run.

  --mode infer (default): clips sharded across ranks, no data-path collective (inference shares nothing); the only collectives
                          are the timing barrier and the MAX over ranks of the elapsed time.  Weak scaling.
  --mode train          : BASELINE metric (ii), train step time: OnsetsFrames fwd + bwd + Adam on `--clips` (default 8) clips x
                          625 frames per GPU, clip-level data parallelism (amt_tools_amd.dp.DataParallelOptimizer: ONE flat RCCL
                          gradient all-reduce inside optimizer.step(), amt_tools/train.py:126-141 unchanged).  Weak scaling.

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline     -- dominant kernel of the timed region, timed live with HIP events on the launch stream
  cpu_baseline -- the oracle (numpy front-end + torch-CPU fp32 model restatement) on a bounded sample
"""
import argparse
import ctypes as C
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402   (importing torch does not initialise the GPU; nothing below does before launch())

SR, HOP, N_MELS, N_FFT = 22050, 512, 229, 2048
CLIP_SAMPLES, CLIP_FRAMES = 319999, 625
DISTINCT_CLIPS = 64                  # distinct synthetic clips per rank (seeds 1234 + 64 rank + i); larger batches tile them
PEAK_MFMA_BF16_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MODEL_FLOPS_PER_FRAME = 26.70e6     # the reference's layer-by-layer arithmetic
# what the engine executes: the pitch head's fc1 (3648 -> 512) and LogisticBank (512 -> 88) are two Linear layers with nothing in
# between in eval mode and are folded into one 3648 -> 88 layer when the weights are loaded (same results, 3.18 MFLOP less per frame)
EXECUTED_FLOPS_PER_FRAME = MODEL_FLOPS_PER_FRAME - 2 * (3648 * 512 + 512 * 88 - 3648 * 88)
TRAIN_FLOPS_PER_FRAME = 80.0e6      # ~3x forward

# algorithmic work per clip-frame; flops for MFMA stages, bytes for streaming stages
STAGE_FLOPS = {
    'conv1': 2 * 2 * 229 * 32 * 9,
    'conv2_pool': 2 * 2 * 229 * 32 * 32 * 9,
    'conv3_pool': 2 * 2 * 114 * 32 * 64 * 9,
    'conv_stack': 2 * 2 * (229 * 32 * 9 + 229 * 32 * 32 * 9 + 114 * 32 * 64 * 9),   # conv1 + conv2 + conv3 as ONE kernel (csrc/convf.hip)
    'fc1_gemm': 2 * 3648 * 512,                 # the onset head's fc1; the pitch head's is folded into its output layer, see below
    'rec_xproj_gemm': 2 * 512 * 1024,
    'rec_bilstm': 2 * 2 * 512 * 128,
    'rec_head_gemm': 2 * 256 * 88,
    'pitch_head_gemm': 2 * 3648 * 88,           # (fc1 . LogisticBank) of the pitch head as ONE K = 3648, N = 88 layer (weights folded at load time)
    'adj_xproj_gemm': 2 * 176 * 1024,
    'adj_bilstm': 2 * 2 * 512 * 128,
    'adj_head_gemm': 2 * 256 * 88,
}
# compulsory HBM bytes per clip-frame of the MFMA stages (bf16 activations; weights are L2-resident)
ALGO_BYTES = {
    'conv2_pool': 2 * (229 * 4 + 114 * 32 * 2),         # both heads: read the log-mel row, write the pooled 114 x 32 map
    'conv3_pool': 2 * (114 * 32 * 2 + 57 * 64 * 2),
    'conv_stack': 2 * (229 * 4 + 57 * 64 * 2),          # both heads: read the log-mel row, write the twice-pooled 57 x 64 map
    'fc1_gemm': 3648 * 2 + 512 * 2,
    'pitch_head_gemm': 3648 * 2 + 88 * 4,
}
STAGE_BYTES = {
    'spec_power': HOP * 4 + N_MELS * 4,            # read hop samples, write the mel-power row
    'spec_scale': N_MELS * 4 * 2,                  # read mel power, write scaled features
    'pianoroll': 2 * 88 * 4 * 2,                   # read 2x88 logits, write 2x88 outputs
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'])
    ap.add_argument('--clips', type=int, default=None, help='clips per GPU per step (default 1024 for infer -- sweep on MI355X, round 3: 512: 37.7, '
                                                            '768: 38.6, 1024: 40.2, 1536: 39.5, 2048: 40.4 M frames/s -- and 8 for train, the reference batch)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f16', 'x3'])
    ap.add_argument('--of2', action='store_true', help='train mode: OnsetsFrames2 as shipped (model_complexity 3, offset head)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='wall-time budget of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--no-train-probe', action='store_true', help='skip the one-GPU training-step time (BASELINE metric ii) appended to the default line')
    ap.add_argument('--no-hcqt', action='store_true', help='skip the BASELINE config-3 leg (OnsetsFrames + HCQT 6 x 72) appended to the default line')
    ap.add_argument('--no-parity', action='store_true', help='skip the bf16-vs-x3 / oracle cell-mismatch count and the x3 throughput leg')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only for the "
                                                      "single-GPU smoke test of the multi-process path)")
    ap.add_argument('--share-device', action='store_true', help='test only: every rank uses cuda:0')
    ap.add_argument('--force-dist', action='store_true', help='create the process group (default backend nccl = RCCL) and run every collective of the '
                                                              'N > 1 path -- timing barrier, MAX / gather of the elapsed times and, in train mode, the '
                                                              'flat gradient all-reduce inside optimizer.step() -- also at --gpus 1: RCCL under test on a one-GPU box')
    ap.add_argument('--dump', default=None, help='test only: directory; every rank writes what it computed (infer: its piano rolls per clip, train: its '
                                                 'weights after the last step) as rank<r>.npz -- tests/test_gpu_multirank.py compares ranks with a one-process run')
    ap.add_argument('--dropout-off', action='store_true', help='test only (train): Dropout p = 0, so that runs of different world sizes are comparable')
    ap.add_argument('--fail-rank', type=int, default=-1, help='test only (with --dry-run): this rank exits with an error before the rendezvous')
    ap.add_argument('--dry-run', action='store_true', help='test only (CPU): ranks rendezvous over gloo and rank 0 prints a line '
                                                           'without touching a GPU -- exercises the launcher and the relay')
    args = ap.parse_args(argv)
    if args.clips is None:
        args.clips = 1024 if args.mode == 'infer' else 8
    return args


# ------------------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ------------------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Start `args.gpus` fresh rank processes of this script and relay rank 0's JSON line.  Runs before anything in this process
    has touched the GPU (children are started with Popen, nothing is exec'ed from a GPU-initialised process)."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's pipe is drained on a thread while every child is polled: one rank dying before or at the rendezvous (out of memory, bad
    # device) must end the run at once -- its peers would otherwise sit in init_process_group / barrier until the store times out
    import threading
    got = {'line': None}

    def drain():
        for out in procs[0].stdout:
            out = out.rstrip('\n')
            if out.startswith('{') and '"metric"' in out:
                got['line'] = out
            elif out:
                print(out, file=sys.stderr, flush=True)

    th = threading.Thread(target=drain, daemon=True)
    th.start()
    deadline = time.time() + float(os.environ.get('AMTX_BENCH_LAUNCH_TIMEOUT', '3000'))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = 'a rank exited with an error'
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() > deadline:
            failed = 'timeout'
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    th.join(timeout=5)
    line = got['line']
    if failed or any(rcs) or line is None:
        print(f'bench.py launcher: {failed or "failure"}: rank exit codes {rcs}, result line {"present" if line else "missing"}',
              file=sys.stderr, flush=True)
        return 1
    print(line, flush=True)
    return 0


def init_ranks(args):
    """(rank, world, device) from the torchrun-style environment; WORLD_SIZE must equal --gpus."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs')
    import torch.distributed as dist
    if args.dry_run:
        if world > 1 or args.force_dist:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            dist.init_process_group('gloo', rank=rank, world_size=world)
        return rank, world, 'cpu'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()) if world == 1 else '29500')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            import tempfile
            base = os.path.join(tempfile.gettempdir(), f'amtx_rccl_{os.getpid()}')
            if rank == 0 and os.environ.get('NCCL_DEBUG', '').upper() in ('', 'VERSION', 'WARN'):
                # which algorithm / protocol RCCL picks for the 19.4 MB gradient all-reduce (ring vs tree vs direct) decides
                # whether a hand-rolled exchange is worth writing: rank 0 logs RCCL's tuning decisions into a file that rccl_choices() parses
                # into the result line, so the first multi-GPU run answers the question without a second one
                _RCCL_LOG['path'] = base + '.log'
                os.environ.update(NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,TUNING', NCCL_DEBUG_FILE=_RCCL_LOG['path'])
            # RCCL prints its version banner with printf, i.e. to this process's STDOUT, whenever NCCL_DEBUG is VERSION or higher (the GPU boxes
            # of this pool export it: the banner showed up BEHIND the result line).  The contract is ONE JSON line there: file descriptor 1
            # points at a side file until main() has torn the group down, then comes back for the result line
            _RCCL_LOG['stdout_path'] = base + '.stdout'
            sys.stdout.flush()
            _RCCL_LOG['saved_stdout'] = os.dup(1)
            fd = os.open(_RCCL_LOG['stdout_path'], os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            os.dup2(fd, 1)
            os.close(fd)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus
    return rank, world, device


_RCCL_LOG = {'path': None, 'stdout_path': None, 'saved_stdout': None}


def restore_stdout():
    """Undo init_ranks' redirection of file descriptor 1 (C stdio flushed into the side file first)."""
    saved = _RCCL_LOG.get('saved_stdout')
    if saved is None:
        return
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:                                            # noqa: BLE001
        pass
    os.dup2(saved, 1)
    os.close(saved)
    _RCCL_LOG['saved_stdout'] = None


def rccl_choices(nbytes=None, limit=12):
    """What rank 0's RCCL logged (NCCL_DEBUG=INFO, subsystems INIT + TUNING): version, channel / topology summary lines and the
    '<bytes> Bytes -> Algo <a> proto <p>' decisions (all sizes seen; `for_allreduce_bytes` = the ones of the gradient all-reduce's size)."""
    import re
    path = _RCCL_LOG.get('path')
    if not path or not os.path.exists(path):
        return None
    spath = _RCCL_LOG.get('stdout_path') or ''
    algo, init = {}, []
    try:
        C.CDLL(None).fflush(None)                       # RCCL's printf output sits in C stdio's buffer while stdout is a file
    except Exception:                                            # noqa: BLE001
        pass
    try:
        with open(path, errors='replace') as f:
            for line in f:
                m = re.search(r'(\d+) Bytes -> Algo (\S+) proto (\S+)(.*)', line)
                if m:
                    key = (int(m.group(1)), m.group(2), m.group(3))
                    algo[key] = algo.get(key, 0) + 1
                elif re.search(r'(RCCL|NCCL) version|Channel|Trees|Rings|comm 0x\S+ rank|xgmi|XGMI|nranks', line) and len(init) < limit:
                    init.append(line.strip()[-200:])
    except OSError:
        return None
    try:
        with open(spath, errors='replace') as f:
            init = [l.strip()[-200:] for l in f if re.search(r'(RCCL|HIP|ROCm) version', l)][:3] + init
    except OSError:
        pass
    rows = [{'bytes': k[0], 'algo': k[1], 'proto': k[2], 'times': n} for k, n in sorted(algo.items())]
    rec = {'log': path, 'env': 'NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,TUNING (rank 0)', 'decisions': rows[:limit * 2], 'init_lines': init}
    if nbytes is not None:
        rec['for_allreduce_bytes'] = [r for r in rows if r['bytes'] == nbytes]
    return rec


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(elapsed, world, device, backend):
    if not _dist_on():
        return elapsed
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(elapsed, world, device, backend):
    """Every rank's own elapsed time (list, rank order): the first SCALE record is then diagnosable without a second run."""
    if not _dist_on():
        return [elapsed]
    import torch.distributed as dist
    t = torch.zeros(world, dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    t[dist.get_rank()] = elapsed
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def barrier(world):
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


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


# ------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = checker; here it is the thing timed, on the host cores, rank 0 at N = 1 only)
# ------------------------------------------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def sustained_matrix_rate():
    """(TFLOP/s, source) of the committed microbenchmark tools/mfma_sustained.py: the bf16 MFMA rate this chip sustains on random operands out
    of registers (its clock settles to the power budget, well under the 2.4 GHz the nominal 2.5 PFLOP/s assumes); None when the file is absent."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'mfma_sustained.json')) as f:
            rec = json.load(f)
        return float(rec['sustained_bf16_tflops_random']), 'profiles/mfma_sustained.json (tools/mfma_sustained.py: register-resident v_mfma stream, random operands)'
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(budget_s, sd, keep=8):
    """Oracle on the host cores: numpy front-end restatement + torch-CPU fp32 model restatement (its recurrences through ATen's
    nn.LSTM, what the reference itself runs on a CPU), one clip per call (the reference's run_offline pattern,
    amt_tools/inference.py:38-41).  Bounded by wall time: clips are processed until `budget_s` is used up (at least one, at most
    64); the clips are synthesised BEFORE the clock starts.  The thread count is capped at 16: the per-step LSTM matmuls are tiny
    and more threads only add synchronisation cost.  Returns (record, piano rolls of the first `keep` clips) -- the latter feed
    the bf16 cell-mismatch count."""
    from amt_tools_amd.synth import synth_clip
    from oracle import frontend_np as fe, model_ref
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    model_ref.LSTM_IMPL = 'aten'
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    # one clip takes ~0.15-0.3 s: synthesise as many as the budget can use, outside the timed loop
    clips = [synth_clip(i) for i in range(8)]
    rolls = []
    t_fe, t_total, n = 0.0, 0.0, 0
    while n < 64:
        if n >= len(clips):
            clips.append(synth_clip(n))
        y = clips[n]
        a = time.perf_counter()
        feats = fe.melspec_process_audio(y, SR, HOP, N_MELS, N_FFT, dtype=np.float32).astype(np.float32)
        b = time.perf_counter()
        with torch.no_grad():
            o = model_ref.run_on_batch(torch.from_numpy(feats[None]), sdt)
        c = time.perf_counter()
        t_fe += b - a
        t_total += c - a
        if n < keep:
            rolls.append((o['onsets'][0].numpy().copy(), o['multi_pitch'][0].numpy().copy(),
                          o['logits']['onsets'][0].numpy().copy(), o['logits']['multi_pitch'][0].numpy().copy()))
        n += 1
        if t_total > budget_s:
            break
    rec = {'value': n * CLIP_FRAMES / t_total, 'unit': 'frames/s', 'cores': cores, 'cores_available': os.cpu_count(), 'cpu_model': cpu_model_name(),
           'kind': 'port',
           'sample': f'{n} synthetic clips x {CLIP_FRAMES} frames in {t_total:.1f} s of oracle time (clip synthesis excluded), one clip per '
                     f'call, fp32, numpy front-end (share {t_fe / t_total:.2f}) + torch-CPU model restatement with ATen nn.LSTM recurrences',
           # the front-end / model split, and the batch-of-8 call next to the one-clip-per-call pattern
           'frontend_frames_per_s': n * CLIP_FRAMES / max(t_fe, 1e-9), 'model_frames_per_s': n * CLIP_FRAMES / max(t_total - t_fe, 1e-9),
           'frontend_share': t_fe / t_total}
    nb = min(8, n)
    if nb >= 2:
        fb = np.stack([fe.melspec_process_audio(clips[i], SR, HOP, N_MELS, N_FFT, dtype=np.float32) for i in range(nb)]).astype(np.float32)
        a = time.perf_counter()
        with torch.no_grad():
            model_ref.run_on_batch(torch.from_numpy(fb), sdt)
        rec['model_frames_per_s_batch8'] = nb * CLIP_FRAMES / (time.perf_counter() - a)
        rec['batch8_note'] = f'{nb} clips in ONE call of the model restatement (front-end excluded)'
    return rec, rolls


def cpu_train_baseline(budget_s):
    """Train-mode CPU reference point: the package's CPU path (stock torch ops on the reference's module layout = the reference's
    own arithmetic) for one fwd + bwd + Adam step on 2 clips x 625 frames of precomputed oracle features."""
    from amt_tools_amd import tools
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd.synth import synth_clip, synth_labels
    from oracle import frontend_np as fe
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = OnsetsFrames(N_MELS, tools.PianoProfile(), 1, 2, device='cpu')
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=6e-4)
    Bc = 2
    feats = np.stack([fe.melspec_process_audio(synth_clip(i), SR, HOP, N_MELS, N_FFT, dtype=np.float32) for i in range(Bc)]).astype(np.float32)
    lab = [synth_labels(i) for i in range(Bc)]
    batch = {tools.KEY_FEATS: torch.from_numpy(feats), tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])),
             tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab]))}
    times = []
    t0 = time.perf_counter()
    while len(times) < 4 and (time.perf_counter() - t0 < budget_s or len(times) < 2):
        a = time.perf_counter()
        opt.zero_grad()
        model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
        opt.step()
        times.append(time.perf_counter() - a)
    step_s = min(times[1:]) if len(times) > 1 else times[0]
    # `value` is what was measured (a 2-clip step); the linear extrapolation to the 8-clip batch over-states the CPU time (the LSTM time
    # steps amortise over the batch) and lives in its own, clearly named field
    return {'value': step_s * 1e3, 'unit': f'ms/step at {Bc} clips per step', 'cores': cores, 'cores_available': os.cpu_count(), 'cpu_model': cpu_model_name(),
            'kind': 'port', 'clips_per_step': Bc,
            'extrapolated_8clip_ms': step_s * 1e3 * 8 / Bc,
            'sample': f'torch-CPU fp32 fwd+bwd+Adam on {Bc} clips x {CLIP_FRAMES} frames of oracle features: {step_s:.2f} s per step '
                      f'(best of {max(1, len(times) - 1)} after one warm-up step); extrapolated_8clip_ms = x{8 // Bc}, an upper bound'}


# ------------------------------------------------------------------------------------------------------------------------------
# inference mode
# ------------------------------------------------------------------------------------------------------------------------------
def parity_leg(model, audio, out_bf16, device, oracle_rolls):
    """After the timed region: (1) the fp32-class `x3` mode on the same clips -- its throughput, and the fraction of piano-roll
    cells in which the headline bf16 mode differs from it; (2) both modes against the CPU oracle's piano rolls of the same clips
    (computed by the cpu_baseline leg).  SURVEY F8: on thresholded outputs the mismatch count IS the parity metric."""
    from amt_tools_amd import tools
    res = {}
    # the mode that is INSIDE north_star's 1e-4 runs the headline batch itself (round 5: two-plane activations, DMA GEMMs, convx.hip) with
    # its own stage timers and roofline block: three bf16 MFMAs per fp32-class product -> its matrix roof is 2.5 PFLOP/s / 3
    Bx = audio.shape[0]
    mx, _, _ = build_model(device, 'x3')
    bx = {tools.KEY_AUDIO: audio[:Bx]}
    from amt_tools_amd import _lib
    L = _lib.lib()
    with torch.no_grad():
        ox = mx.run_on_batch(bx)
        ox = mx.run_on_batch(bx)
        torch.cuda.synchronize()
        engx = mx._get_engine(torch.device(device))
        _lib.check(L.amtx_of_profile_enable(engx.handle, 1))
        nx = 5
        t0 = time.perf_counter()
        for _ in range(nx):
            ox = mx.run_on_batch(bx)
        torch.cuda.synchronize()
        dtx = (time.perf_counter() - t0) / nx
        stage_ms = (C.c_double * L.amtx_of_num_stages())()
        nfw = C.c_int(0)
        _lib.check(L.amtx_of_profile_read(engx.handle, stage_ms, C.byref(nfw)))
        _lib.check(L.amtx_of_profile_enable(engx.handle, 0))
    res['x3_frames_per_s'] = Bx * CLIP_FRAMES / dtx
    res['x3_clips_per_step'] = Bx
    plx = {L.amtx_of_stage_name(i).decode(): stage_ms[i] / max(1, nfw.value) for i in range(L.amtx_of_num_stages())}
    domx = max((k for k in plx if k in STAGE_FLOPS), key=plx.get)
    peak3 = PEAK_MFMA_BF16_TFLOPS / 3.0
    achx = STAGE_FLOPS[domx] * Bx * CLIP_FRAMES / (plx[domx] * 1e-3) / 1e12
    res['x3'] = {
        'frames_per_s': res['x3_frames_per_s'], 'ms_per_step': dtx * 1e3, 'clips_per_step': Bx, 'dtype': 'bf16x3',
        'whole_path_frac_of_its_mfma_roof': res['x3_frames_per_s'] * MODEL_FLOPS_PER_FRAME / (2.5e15 / 3.0),
        'roofline': {'kernel': domx, 'bound': 'mfma', 'achieved': achx, 'peak': peak3, 'unit': 'TFLOP/s', 'frac': achx / peak3, 'traffic': None,
                     'avg_launch_ms': plx[domx], 'kernel_ms_per_step': {k: round(v, 4) for k, v in sorted(plx.items(), key=lambda kv: -kv[1])},
                     'note': 'achieved = fp32-class algorithmic flops of the stage (SURVEY 8d) / its HIP-event time (front-end kernels not in this '
                             'table); peak = dense bf16 MFMA peak / 3: every product is hi.hi + hi.lo + lo.hi'},
    }
    nd = min(8, Bx)       # clips compared with the x3 mode / the CPU oracle (the oracle leg keeps the first 8 of the 64 distinct clips)
    # every clip of the batch against the distinct clip it is a copy of: a block- or tail-dependent indexing error in any kernel of the
    # 1024-clip run would show here (clips 0 .. 7 alone are block 0 of most grids)
    Bfull = out_bf16[tools.KEY_ONSETS].shape[0]
    same = True
    for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        o = out_bf16[k]
        idx = torch.arange(Bfull, device=o.device) % DISTINCT_CLIPS
        same = same and bool(torch.equal(o, o[idx]))
    res['tiled_clips_equal_their_source_clip'] = same
    res['tiled_clips_compared'] = Bfull
    cells, diff = 0, 0
    for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        a, b = out_bf16[k][:nd], ox[k][:nd]
        cells += a.numel()
        diff += int((a != b).sum().item())
    res['bf16_cell_mismatch_rate_vs_x3'] = diff / cells
    res['bf16_cells_compared'] = cells
    if oracle_rolls:
        n = min(nd, len(oracle_rolls))
        for tag, o in (('bf16', out_bf16), ('x3', ox)):
            d = c = 0
            for i in range(n):
                for j, k in enumerate((tools.KEY_ONSETS, tools.KEY_MULTIPITCH)):
                    got = o[k][i].cpu().numpy()
                    d += int((got != oracle_rolls[i][j]).sum())
                    c += got.size
            res[f'{tag}_cell_mismatch_rate_vs_cpu_oracle'] = d / c
        res['oracle_clips_compared'] = n
        # x3 logits against the oracle's: the 1e-4 gate of north_star, on the bench workload itself
        lx = mx.engine_logits(mx.frontend(audio[:n].unsqueeze(-2)))
        err = 0.0
        for i in range(n):
            err = max(err, float(np.abs(lx['onsets'][i].cpu().numpy() - oracle_rolls[i][2]).max()),
                      float(np.abs(lx['multi_pitch'][i].cpu().numpy() - oracle_rolls[i][3]).max()))
        res['x3_max_abs_logit_err_vs_cpu_oracle'] = err
    del mx
    torch.cuda.empty_cache()
    # precision 'f16' (optional build, AMTX_BUILD_F16=1): the headline kernels with IEEE half operands, at the headline batch
    if not L.amtx_has_f16():
        if oracle_rolls:
            n = min(nd, len(oracle_rolls))
            lx = model.engine_logits(model.frontend(audio[:n].unsqueeze(-2)))
            e = 0.0
            for i in range(n):
                e = max(e, float(np.abs(lx['onsets'][i].cpu().numpy() - oracle_rolls[i][2]).max()),
                        float(np.abs(lx['multi_pitch'][i].cpu().numpy() - oracle_rolls[i][3]).max()))
            res['bf16_max_abs_logit_err_vs_cpu_oracle'] = e
            res['precision_modes'] = [
                {'mode': 'bf16 (headline; opt-in: precision="bf16")', 'frames_per_s': None, 'cell_mismatch_rate_vs_cpu_oracle': res.get('bf16_cell_mismatch_rate_vs_cpu_oracle'),
                 'max_abs_logit_err_vs_cpu_oracle': e},
                {'mode': 'x3 (meets 1e-4; the default of the drop-in classes)', 'clips_per_step': res['x3_clips_per_step'], 'frames_per_s': res['x3_frames_per_s'],
                 'cell_mismatch_rate_vs_cpu_oracle': res.get('x3_cell_mismatch_rate_vs_cpu_oracle'),
                 'max_abs_logit_err_vs_cpu_oracle': res.get('x3_max_abs_logit_err_vs_cpu_oracle')}]
        return res
    mf, _, _ = build_model(device, 'f16')
    bf = {tools.KEY_AUDIO: audio}
    with torch.no_grad():
        of = mf.run_on_batch(bf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            of = mf.run_on_batch(bf)
        torch.cuda.synchronize()
        res['f16_frames_per_s'] = 3 * audio.shape[0] * CLIP_FRAMES / (time.perf_counter() - t0)
    if oracle_rolls:
        n = min(nd, len(oracle_rolls))
        d = c = 0
        for i in range(n):
            for j, k in enumerate((tools.KEY_ONSETS, tools.KEY_MULTIPITCH)):
                got = of[k][i].cpu().numpy()
                d += int((got != oracle_rolls[i][j]).sum())
                c += got.size
        res['f16_cell_mismatch_rate_vs_cpu_oracle'] = d / c
        errs = {}
        for tag, mm in (('f16', mf), ('bf16', model)):
            lx = mm.engine_logits(mm.frontend(audio[:n].unsqueeze(-2)))
            e = 0.0
            for i in range(n):
                e = max(e, float(np.abs(lx['onsets'][i].cpu().numpy() - oracle_rolls[i][2]).max()),
                        float(np.abs(lx['multi_pitch'][i].cpu().numpy() - oracle_rolls[i][3]).max()))
            errs[tag] = e
            res[f'{tag}_max_abs_logit_err_vs_cpu_oracle'] = e
        # the precision / throughput trade-off in one place: north_star asks for 1e-4 on the activations
        res['precision_modes'] = [
            {'mode': 'bf16 (headline)', 'frames_per_s': None, 'cell_mismatch_rate_vs_cpu_oracle': res.get('bf16_cell_mismatch_rate_vs_cpu_oracle'),
             'max_abs_logit_err_vs_cpu_oracle': errs['bf16']},
            {'mode': 'f16', 'frames_per_s': res['f16_frames_per_s'], 'cell_mismatch_rate_vs_cpu_oracle': res['f16_cell_mismatch_rate_vs_cpu_oracle'],
             'max_abs_logit_err_vs_cpu_oracle': errs['f16']},
            {'mode': 'x3 (meets 1e-4)', 'clips_per_step': res['x3_clips_per_step'], 'frames_per_s': res['x3_frames_per_s'], 'cell_mismatch_rate_vs_cpu_oracle': res.get('x3_cell_mismatch_rate_vs_cpu_oracle'),
             'max_abs_logit_err_vs_cpu_oracle': res.get('x3_max_abs_logit_err_vs_cpu_oracle')}]
    del mf
    return res


def hcqt_leg(device, clips=512, steps=5):
    """BASELINE config 3 after the timed region: audio resident in HBM -> HIP HCQT (6 harmonics x 72 bins, amt_tools/features/hvqt.py:107-133)
    as model.frontend -> OnsetsFrames(dim_in 72, 6 channels, mc 2), bf16, the same synthetic clips.  Returns frames/s + ms per step."""
    from amt_tools_amd import tools
    from amt_tools_amd.features import HCQT
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd.synth import synth_clip, synth_state_dict
    mod = HCQT(sample_rate=SR, hop_length=HOP, n_bins=72, bins_per_octave=12, device=device)
    model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device=device, precision='bf16')
    sd = synth_state_dict(0, dim_in=72, in_channels=6, model_complexity=2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.frontend = torch.nn.Sequential(mod.frontend())
    model.change_device()
    model.eval()
    base = np.stack([synth_clip(i) for i in range(8)])
    audio = torch.from_numpy(base).to(device).repeat((clips + 7) // 8, 1)[:clips].contiguous()
    with torch.no_grad():
        for _ in range(2):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        # the front-end alone, in the form run_on_batch uses it: (B,T,F,8) bf16 in the first conv's staging format when the engine takes that
        feats16 = bool(model._get_engine(torch.device(device)).takes_feats16())
        (mod.process_batch16 if feats16 else mod.process_batch)(audio)      # untimed: first call of this form (allocations)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            if feats16:
                mod.process_batch16(audio)
            else:
                model.frontend(audio[:, None, :])
        torch.cuda.synchronize()
        fe_ms = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # the engine's stages by its own HIP events (a second, untimed pair of passes), for the leg's own roofline blocks
        from amt_tools_amd import _lib
        L = _lib.lib()
        eng = model._get_engine(torch.device(device))
        _lib.check(L.amtx_of_profile_enable(eng.handle, 1))
        for _ in range(2):
            model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        sms = (C.c_double * L.amtx_of_num_stages())()
        nfw = C.c_int(0)
        _lib.check(L.amtx_of_profile_read(eng.handle, sms, C.byref(nfw)))
        _lib.check(L.amtx_of_profile_enable(eng.handle, 0))
        stages = {L.amtx_of_stage_name(i).decode(): sms[i] / max(1, nfw.value) for i in range(L.amtx_of_num_stages()) if sms[i] > 0}
    T = out[tools.KEY_ONSETS].shape[-1]
    fps = clips * T / dt
    del model, out, audio
    torch.cuda.empty_cache()
    # per-kernel rooflines of the leg.  conv1 + conv2 of the two heads (one stage, convg.hip's fused kernel): 2 heads x 2 x
    # (6 x 9 x 32 + 32 x 9 x 32) x 72 bins = 3.15 MFLOP per frame.  Front-end: audio in (512 samples x 4 B per frame) + the feature map out
    # (72 bins x 16 B in the staging format, 6 x 72 x 4 B otherwise) against HBM
    conv_ms = stages.get('conv2_pool', 0.0) + stages.get('conv1', 0.0)
    conv_flops = 2 * 2 * (6 * 9 * 32 + 32 * 9 * 32) * 72 * clips * T
    fe_bytes = (HOP * 4 + (72 * 16 if feats16 else 6 * 72 * 4)) * clips * T
    rl = {'conv1_conv2': {'kernel': 'conv3x3_gen_kernel (fused first conv + conv2, both heads)', 'bound': 'mfma', 'achieved': conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                          'peak': PEAK_MFMA_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS if conv_ms else None,
                          'avg_launch_ms': conv_ms, 'algorithmic_flops': conv_flops},
          'frontend': {'kernel': 'HCQT front-end (7 decimations, basis products, scaling)', 'bound': 'hbm', 'achieved': fe_bytes / (fe_ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                       'frac': fe_bytes / (fe_ms * 1e-3) / 1e9 / 8000.0, 'ms': fe_ms, 'algorithmic_bytes': fe_bytes,
                       'note': 'algorithmic bytes = audio in + feature map out; the pyramid levels and the power map in between are the traffic on top (profiles/r05ze_hcqt_pmc.txt)'}}
    # the front-end's HBM traffic from the committed PMC passes (profiles/pmc_traffic_hcqt.json, tools/pmc_hcqt_traffic.py), scaled to this batch
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_hcqt.json')) as f:
            pmc = json.load(f)
        rl['frontend']['traffic'] = pmc['frontend_hbm_bytes_per_pass'] * clips / pmc['clips']
        rl['frontend']['traffic_unit'] = 'bytes per front-end pass'
        rl['frontend']['traffic_over_algorithmic'] = rl['frontend']['traffic'] / fe_bytes
        rl['frontend']['traffic_source'] = pmc.get('_source')
        rl['frontend']['traffic_by_kernel'] = {k: v['hbm_bytes_per_pass'] * clips / pmc['clips'] for k, v in pmc.items() if isinstance(v, dict)}
    except (OSError, ValueError, KeyError):
        rl['frontend']['traffic'] = None
    # OF1 + HCQT(6 x 72) = 10.3 MFLOP per frame
    return {'frames_per_s': fps, 'ms_per_step': dt * 1e3, 'clips_per_step': clips, 'frames_per_clip': int(T), 'frontend_ms_per_step': fe_ms,
            'features': '(B,T,F,8) bf16, amtx_cqt_forward16 -> amtx_of_forward_feats16' if feats16 else '(B,C,F,T) fp32',
            'engine_stage_ms': stages, 'roofline': rl,
            'frac_of_mfma_roof': fps * 10.3e6 / 2.5e15,
            'workload': 'BASELINE config 3: OnsetsFrames(mc=2, dim_in 72, 6 channels) + HCQT(6 harmonics x 72 bins, hop 512) inference, bf16, '
                        'audio resident in HBM -> piano rolls'}


def hcqt_x3_leg(device, clips=512, steps=3):
    """BASELINE config 3 in the engine precision that is inside north_star's 1e-4 (x3: split-bf16, three MFMAs per product): the same audio ->
    HCQT -> OnsetsFrames pass as hcqt_leg; since round 6 the features travel as the two 16-bit planes of the split ((2,B,T,F,8),
    amtx_cqt_forward16_split) and conv1 + conv2 run on convx.hip's layer-specialised kernel."""
    from amt_tools_amd import tools
    from amt_tools_amd.features import HCQT
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd.synth import synth_clip, synth_state_dict
    mod = HCQT(sample_rate=SR, hop_length=HOP, n_bins=72, bins_per_octave=12, device=device)
    model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device=device, precision='x3')
    sd = synth_state_dict(0, dim_in=72, in_channels=6, model_complexity=2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.frontend = torch.nn.Sequential(mod.frontend())
    model.change_device()
    model.eval()
    base = np.stack([synth_clip(i) for i in range(8)])
    audio = torch.from_numpy(base).to(device).repeat((clips + 7) // 8, 1)[:clips].contiguous()
    with torch.no_grad():
        for _ in range(2):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        feats16 = int(model._get_engine(torch.device(device)).takes_feats16())
    T = out[tools.KEY_ONSETS].shape[-1]
    del model, out, audio
    torch.cuda.empty_cache()
    return {'frames_per_s': clips * T / dt, 'ms_per_step': dt * 1e3, 'clips_per_step': clips,
            'features': '(2,B,T,F,8) bf16 planes, amtx_cqt_forward16_split -> amtx_of_forward_feats16' if feats16 == 2 else '(B,C,F,T) fp32',
            'note': 'engine precision x3 (inside 1e-4 of the fp32 reference: tests/test_gpu_model.py::test_config3_hcqt_frontend_fused_into_the_model)'}


def run_infer(args, rank, world, device):
    from amt_tools_amd import _lib, tools
    from amt_tools_amd.synth import synth_clip

    model, mel, sd = build_model(device, args.precision)
    B = args.clips
    # clips are dealt round-robin over ranks (clip g belongs to rank g % world): rank r's i-th distinct clip is clip i world + r
    base = np.stack([synth_clip(i * world + rank) for i in range(DISTINCT_CLIPS)])
    audio = torch.from_numpy(base).to(device).repeat((B + DISTINCT_CLIPS - 1) // DISTINCT_CLIPS, 1)[:B].contiguous()
    batch = {tools.KEY_AUDIO: audio}

    def step():
        with torch.no_grad():
            return model.run_on_batch(batch)

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
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier(world)
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_rank = gather_over_ranks(elapsed, world, device, args.backend)
    elapsed = max_over_ranks(elapsed, world, device, args.backend)
    if args.dump:
        os.makedirs(args.dump, exist_ok=True)
        nd = min(B, DISTINCT_CLIPS)
        np.savez(os.path.join(args.dump, f'rank{rank}.npz'), onsets=out[tools.KEY_ONSETS][:nd].cpu().numpy().astype(np.uint8),
                 multi_pitch=out[tools.KEY_MULTIPITCH][:nd].cpu().numpy().astype(np.uint8), clip_ids=np.arange(nd) * world + rank)

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

    if eng.conv_stack_fused(B, CLIP_FRAMES):
        # the engine's stage timer books the fused convolution kernel under conv2_pool and launches nothing for conv3_pool
        per_launch['conv_stack'] = per_launch.pop('conv2_pool')
        per_launch.pop('conv3_pool', None)
    if rank != 0:
        return None
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
    # HBM traffic from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes, FETCH_SIZE doubled per MI355X_MICROARCH.md -- counters and timing cannot share a run), scaled to this launch's clips
    hbm_measured = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc = json.load(f)
        src = pmc.get('_source', 'profiles/pmc_traffic.json')
        kernels = {k: v for k, v in pmc.items() if isinstance(v, dict) and 'hbm_bytes_corrected' in v}
        # the dominant stage's own kernel (entries are per kernel; a kernel that serves several stages cannot be attributed to one)
        own = [v for v in kernels.values() if v.get('stages') == [dom]]
        if own:
            roof['traffic'] = own[0]['hbm_bytes_corrected'] * B / own[0]['clips']
            roof['traffic_unit'] = 'bytes per launch'
            roof['traffic_source'] = src
            roof['algorithmic_bytes'] = ALGO_BYTES.get(dom, 0) * frames_per_launch or None
        # every kernel of a step (tools/pmc_traffic.py), launches per step included
        tot = sum(v['hbm_bytes_corrected'] * v.get('launches_per_step', 1) / v['clips'] for v in kernels.values())
        hbm_measured = (tot * B, src)
    except (OSError, ValueError, KeyError):
        pass
    sus = sustained_matrix_rate()
    if sus is not None and roof['bound'] == 'mfma':
        roof['sustained_peak'] = sus[0]
        roof['frac_of_sustained_matrix_rate'] = roof['achieved'] / sus[0]
        roof['sustained_peak_source'] = sus[1]
    roof['avg_launch_ms'] = per_launch[dom]
    roof['kernel_ms_per_step'] = {k: round(v, 4) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])}
    fps = total_frames / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    config = {'workload': 'OnsetsFrames(mc=2)+MelSpec(229 bins, n_fft 2048, hop 512) inference, synthetic 22.05 kHz clips of 319999 '
                          'samples (625 frames; 64 distinct clips per rank -- clip g of the job belongs to rank g % world -- tiled into separate HBM buffers), audio resident in HBM -> piano rolls',
              'clips_per_gpu_per_step': B, 'frames_per_clip': CLIP_FRAMES, 'parallelism': f'clip-sharded x{world}, no collectives',
              'rccl_ranks': world, 'process_group': (args.backend if _dist_on() else None), 'per_rank_frames_per_s': [B * CLIP_FRAMES * args.steps / t for t in per_rank],
              'whole_path_frac_of_mfma_roof': fps / world * MODEL_FLOPS_PER_FRAME / 2.5e15,
              'whole_path_frac_of_mfma_roof_executed_flops': fps / world * EXECUTED_FLOPS_PER_FRAME / 2.5e15,
              'flops_per_frame': {'reference_algorithmic': MODEL_FLOPS_PER_FRAME, 'executed': EXECUTED_FLOPS_PER_FRAME,
                                  'note': 'pitch head fc1 + LogisticBank folded into one linear layer at weight load (eval mode has nothing between them)'},
              'whole_path_frac_of_compulsory_hbm_roof': fps / world * 2752 / 8.0e12}
    if sus is not None:
        config['whole_path_frac_of_sustained_matrix_rate'] = fps / world * MODEL_FLOPS_PER_FRAME / (sus[0] * 1e12)
    if hbm_measured is not None:
        # bytes this build really moves per step (sum of the PMC-counted kernels) over the step time, against 8 TB/s
        config['whole_path_hbm_frac_measured'] = hbm_measured[0] / (ms_per_step * 1e-3) / 8.0e12
        config['whole_path_hbm_bytes_per_frame_measured'] = hbm_measured[0] / frames_per_launch
        config['whole_path_hbm_source'] = hbm_measured[1]
    res = {
        'metric': 'audio frames/sec (OnsetsFrames+Mel-229 inference)', 'value': fps, 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'bf16': 'bf16', 'f16': 'f16', 'x3': 'bf16x3'}[args.precision], 'data': 'synthetic',
        'config': config, 'roofline': roof,
    }
    oracle_rolls = None
    if world == 1 and args.cpu_seconds > 0:
        res['cpu_baseline'], oracle_rolls = cpu_baseline(args.cpu_seconds, sd)
    if world == 1 and not args.no_parity and args.precision == 'bf16':
        config.update(parity_leg(model, audio, out, device, oracle_rolls))
        if config.get('precision_modes'):
            config['precision_modes'][0]['frames_per_s'] = fps
    if world == 1 and not args.no_hcqt and args.precision == 'bf16':
        config['hcqt'] = hcqt_leg(device)
        config['hcqt_frames_per_s'] = config['hcqt']['frames_per_s']
        config['hcqt']['x3'] = hcqt_x3_leg(device)
    if world == 1 and not args.no_train_probe:
        # BASELINE metric (ii), train step time, at N = 1 (the DP = 8 figure needs the 8-GPU node: python bench.py --mode train --gpus 8)
        del model, out
        torch.cuda.empty_cache()
        config['train'] = train_step_probe(device)
        config['train_step_ms_1gpu'] = config['train']['ms_per_step']
        config['train_step_workload'] = 'OnsetsFrames(mc=2)+MelSpec(229) fwd+bwd+Adam, 8 clips x 625 frames per GPU, 10 steps after 3 warm-up steps (python bench.py --mode train)'
        config['train_allreduce_probe'] = train_allreduce_probe()
    return res


# ------------------------------------------------------------------------------------------------------------------------------
# training mode (BASELINE metric ii)
# ------------------------------------------------------------------------------------------------------------------------------
def _train_setup(device, rank, B, of2, dropout_off=False):
    """Model, optimizer and one synthetic labelled batch of the training step (amt_tools/train.py:122-141); returns step()."""
    from amt_tools_amd import tools
    from amt_tools_amd.dp import DataParallelOptimizer, broadcast_parameters
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
    from amt_tools_amd.synth import synth_clip, synth_labels

    torch.manual_seed(0)
    if of2:
        model = OnsetsFrames2(N_MELS, tools.PianoProfile(), 1, device=device)
    else:
        model = OnsetsFrames(N_MELS, tools.PianoProfile(), 1, 2, device=device)
    model.frontend = torch.nn.Sequential(MelSpec(sample_rate=SR, hop_length=HOP, n_mels=N_MELS, n_fft=N_FFT, device=device).frontend())
    model.change_device()
    broadcast_parameters(model)
    if dropout_off:
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    model.train()
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=6e-4, buffers=model.buffers(),   # BatchNorm policy: amt_tools_amd/dp.py
                                force_collective=True if _dist_on() else None)
    audio = torch.from_numpy(np.stack([synth_clip(rank * B + i) for i in range(B)])).to(device)
    lab = [synth_labels(rank * B + i) for i in range(B)]
    batch = {tools.KEY_AUDIO: audio,
             tools.KEY_MULTIPITCH: torch.from_numpy(np.stack([l[0] for l in lab])).to(device),
             tools.KEY_ONSETS: torch.from_numpy(np.stack([l[1] for l in lab])).to(device)}
    if of2:
        batch[tools.KEY_OFFSETS] = torch.from_numpy(np.stack([l[1][:, ::-1].copy() for l in lab])).to(device)   # any sparse binary map

    def step():       # amt_tools/train.py:122-141
        opt.zero_grad()
        loss = model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
        return loss

    step.model = model
    return step, opt


def recurrence_latency_probe(device, B=8, T=CLIP_FRAMES, H=128, reps=5):
    """The training step's dependency chain, measured: the persistent BiLSTM kernels of one recurrence (amtx_bilstm_h_train_fwd / _bwd, 8 clips x
    625 steps, hidden 128, both directions in one launch) timed on their own with HIP events on the launch stream -> microseconds per time step."""
    from amt_tools_amd import _lib
    L = _lib.lib()
    dev = torch.device(device)
    g = torch.Generator(device='cpu').manual_seed(0)
    xproj = (torch.randn(1, B * T, 8 * H, generator=g) * 0.5).to(dev)
    whf, whb = [(torch.randn(4 * H, H, generator=g) * 0.05).to(dev) for _ in range(2)]
    n = int(L.amtx_bilstm_h_packed_elems(H, 2))
    ff = torch.empty((1, n), dtype=torch.int16, device=dev)
    fb = torch.empty((1, n), dtype=torch.int16, device=dev)
    out = torch.empty((1, B, T, 2 * H), dtype=torch.float32, device=dev)
    save = torch.empty((1, B, T, 2, 5, H), dtype=torch.float32, device=dev)
    dout = (torch.randn(1, B, T, 2 * H, generator=g) * 0.1).to(dev)
    dxp = torch.empty_like(xproj)
    st = _lib.current_stream(dev)
    with torch.cuda.device(dev):
        _lib.check(L.amtx_bilstm_h_pack_device(_lib.ptr(whf), _lib.ptr(whb), H, 2, _lib.ptr(ff), _lib.ptr(fb), st), 'amtx_bilstm_h_pack_device')
        res = {}
        for name, call in (('fwd', lambda: L.amtx_bilstm_h_train_fwd(_lib.ptr(xproj), _lib.ptr(ff), H, 2, _lib.ptr(out), _lib.ptr(save), B, T, 1, st)),
                           ('bwd', lambda: L.amtx_bilstm_h_train_bwd(_lib.ptr(dout), _lib.ptr(save), _lib.ptr(fb), H, 2, _lib.ptr(dxp), B, T, 1, st))):
            _lib.check(call(), name)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _lib.check(call(), name)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / reps
    return {'fwd_kernel_ms': res['fwd'], 'bwd_kernel_ms': res['bwd'], 'fwd_us_per_step': res['fwd'] * 1e3 / T, 'bwd_us_per_step': res['bwd'] * 1e3 / T}


def train_step_probe(device, steps=10, warmup=3):
    """BASELINE metric (ii) at N = 1 inside the default line: ms per fwd + bwd + Adam step, 8 clips x 625 frames -- with the step's flops, its
    fraction of the matrix roof and the LATENCY MODEL SURVEY 8(d) asks for: the step is a chain of 4 x 625 dependent recurrence steps (onset head
    and refinement stage, forward and backward), whose per-step time is measured on the kernels themselves."""
    step, _ = _train_setup(device, 0, 8, False)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del step
    frames = 8 * CLIP_FRAMES
    flops = TRAIN_FLOPS_PER_FRAME * frames
    rec = {'ms_per_step': ms, 'clips_per_step': 8, 'frames_per_step': frames, 'flops_per_step': flops, 'achieved_tflops': flops / (ms * 1e-3) / 1e12,
           'frac_of_mfma_roof': flops / (ms * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS,
           'note': 'flops_per_step = 80 MFLOP per clip-frame (SURVEY 8d: ~3x forward) x 5000 frames; the products run as 3 bf16 MFMAs each (split-bf16), '
                   'not counted.  At 8 clips per GPU the step is bound by its dependency chain, not by either roof: see latency_model'}
    sus = sustained_matrix_rate()
    if sus is not None:
        rec['frac_of_sustained_matrix_rate'] = rec['achieved_tflops'] / sus[0]
    try:
        lat = recurrence_latency_probe(device)
        chain = 2 * (lat['fwd_kernel_ms'] + lat['bwd_kernel_ms'])
        lat.update({'dependent_steps': 4 * CLIP_FRAMES, 'recurrence_chain_ms': chain, 'share_of_step': chain / ms,
                    'model': 'step >= 2 x (fwd + bwd recurrence kernel) = 4 x 625 dependent time steps (onset head + refinement stage, each way); each '
                             'recurrence occupies 4 of 256 CUs (2 directions x 8 clips / 4 clips per block), the dense layers between them cannot start '
                             'before their recurrence ends (autograd order), the pitch head overlaps on a side stream'})
        rec['latency_model'] = lat
    except Exception as e:                                      # noqa: BLE001 -- a probe: report, never fail the line
        rec['latency_model'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    return rec


def train_allreduce_probe(timeout=240):
    """The training step's gradient exchange timed somewhere even when the driver has one GPU: `bench.py --mode train
    --force-dist` as a CHILD process (a one-rank RCCL group, every collective of the N-GPU step really issued; a child so that a stuck
    rendezvous can only cost this probe its time limit) -- returns its step time, all-reduce time and bytes, or the reason it has none."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'train', '--force-dist', '--steps', '5', '--warmup', '2', '--cpu-seconds', '0']
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1]
        c = json.loads(line)['config']
        return {'ms_per_step_with_rccl_group': json.loads(line)['ms_per_step'], 'allreduce_ms_per_step': c['allreduce_ms_per_step'],
                'allreduce_bytes': c.get('allreduce_bytes'), 'collectives_per_step': c['collectives_per_step'], 'process_group': c['process_group'],
                'rccl_ranks': c['rccl_ranks'], 'command': 'python bench.py --mode train --force-dist --steps 5 --warmup 2'}
    except Exception as e:                                      # noqa: BLE001 -- a probe: report, never fail the line
        return {'error': f'{type(e).__name__}: {e}'[:200]}


def run_train(args, rank, world, device):
    B = args.clips
    step, opt = _train_setup(device, rank, B, args.of2, args.dropout_off)

    for _ in range(args.warmup):
        loss = step()
    gc.collect()
    gc.disable()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier(world)
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_rank = gather_over_ranks(elapsed, world, device, args.backend)
    elapsed = max_over_ranks(elapsed, world, device, args.backend)
    # the gradient all-reduce on its own (flatten + RCCL all-reduce + unflatten, what DataParallelOptimizer.step() adds to an optimizer step),
    # timed after the run on this rank's last gradients
    allreduce_ms = None
    if _dist_on():
        barrier(world)
        t1 = time.perf_counter()
        for _ in range(5):
            opt.allreduce_gradients()
        barrier(world)
        allreduce_ms = max_over_ranks((time.perf_counter() - t1) / 5 * 1e3, world, device, args.backend)
    if args.dump:
        os.makedirs(args.dump, exist_ok=True)
        torch.cuda.synchronize()
        np.savez(os.path.join(args.dump, f'rank{rank}.npz'), loss=float(loss.detach()),
                 **{k: v.detach().float().cpu().contiguous().numpy() for k, v in step.model.state_dict().items() if not k.startswith('frontend.')})
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    fps = world * B * CLIP_FRAMES * args.steps / elapsed
    name = 'OnsetsFrames2(mc=3)' if args.of2 else 'OnsetsFrames(mc=2)'
    flops = (3 * 93.6e6 if args.of2 else TRAIN_FLOPS_PER_FRAME)
    ach = flops * fps / world / 1e12
    from amt_tools_amd.autograd import training_backend
    res = {
        'metric': f'train step time ({name} fwd+bwd+Adam, DP over clips)', 'value': ms, 'unit': 'ms/step', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (split-bf16 MFMA products in the HIP kernels, fp32 accumulate / parameters / optimizer)', 'data': 'synthetic',
        'config': {'workload': f'{name}+MelSpec(229) training step, {B} clips x {CLIP_FRAMES} frames per GPU, Adam lr 6e-4, labels Bernoulli '
                               f'(synth_labels), audio resident in HBM', 'clips_per_gpu_per_step': B, 'global_batch': world * B,
                   'frames_per_s': fps, 'parallelism': f'dp{world}: one flat fp32 gradient all-reduce per step', 'rccl_ranks': world,
                   'per_rank_ms_per_step': [t / args.steps * 1e3 for t in per_rank], 'allreduce_ms_per_step': allreduce_ms,
                   'allreduce_bytes': (int(opt._flat.numel()) * 4 if getattr(opt, '_flat', None) is not None else None),
                   'collectives_per_step': opt.collectives_run / max(1, args.steps + args.warmup + (5 if allreduce_ms is not None else 0)),
                   'process_group': (args.backend if _dist_on() else None),
                   'rccl': rccl_choices(int(opt._flat.numel()) * 4 if getattr(opt, '_flat', None) is not None else None),
                   'loss': float(loss.detach()), 'backward': training_backend()},
        'roofline': {'kernel': 'whole step', 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_MFMA_BF16_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach / PEAK_MFMA_BF16_TFLOPS, 'traffic': None,
                     'note': f'{flops / 1e6:.1f} MFLOP per clip-frame (SURVEY 8d: ~3x forward) x frames / step time; at 8 clips per GPU the step is '
                             f'latency-bound (4 x 625 dependent recurrence steps), see DESIGN.md 5.4'},
    }
    if world == 1 and args.cpu_seconds > 0 and not args.of2:
        res['cpu_baseline'] = cpu_train_baseline(args.cpu_seconds)
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch(args, argv))            # nothing above has touched the GPU
    if args.dry_run and args.fail_rank >= 0 and int(os.environ.get('RANK', '0')) == args.fail_rank:
        sys.exit(3)
    rank, world, device = init_ranks(args)
    if args.dry_run:
        import torch.distributed as dist
        elapsed = max_over_ranks(1e-3 * (rank + 1), world, 'cpu', 'gloo')
        if _dist_on():
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'dry run (launcher / rendezvous / relay only)', 'value': 0.0, 'unit': 'none', 'n_gpus': world,
                              'steps': 0, 'warmup': 0, 'ms_per_step': elapsed * 1e3, 'dry_run': True,
                              'process_group': 'gloo' if _dist_on() else None}), flush=True)
        if _dist_on():
            dist.destroy_process_group()
        return
    try:
        res = run_infer(args, rank, world, device) if args.mode == 'infer' else run_train(args, rank, world, device)
        if _dist_on():
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
    finally:
        restore_stdout()
    if rank == 0:
        print(json.dumps(res), flush=True)          # the last thing this process writes to stdout


if __name__ == '__main__':
    main()
