"""
Host-side mirror of amt-tools' TranscriptionModel plugin API (amt_tools/models/common.py,
amt_tools/models/onsetsframes.py) for the Onsets & Frames hot path.

Same class names, constructor arguments, attributes (`dim_in, profile, in_channels, model_complexity,
frame_width, device, iter, frontend`), methods (`change_device, pre_proc, forward, post_proc,
run_on_batch, model_name`) and -- because the sub-modules keep the reference's attribute names --
the same `state_dict()` keys, so reference checkpoints load and `amt_tools.train.train()` /
`amt_tools.inference.run_offline()` accept these objects unchanged (seam S2).

Execution paths
---------------
* eval mode on a CUDA (ROCm) device  -> the HIP inference engine behind include/amtx.h
  (`amtx_of_forward`): conv1 -> conv3x3+BN+ReLU+pool (MFMA) -> fc1 (MFMA GEMM) -> persistent BiLSTM ->
  LogisticBank heads -> piano roll.  No fallback: a missing extension raises.
* training mode on a GPU (autograd) -> hand-written HIP forward + backward kernels behind autograd.Functions
  (amt_tools_amd/autograd.py): the three BiLSTM recurrences, BatchNorm(batch statistics)+ReLU+MaxPool, the
  LogisticBank loss; `autograd.training_backend()` names what still runs on ATen.
* CPU devices -> stock torch ops on the same parameters (ATen; the reference's own arithmetic).

`precision` selects the inference engine's dense arithmetic.  `'x3'` (DEFAULT since round 6: split-bf16, three bf16 MFMAs per
product, fp32 accumulate) is the mode whose activations are inside north_star's 1e-4 of the fp32 CPU reference -- it is what a user
who swaps the reference classes for these gets.  `'bf16'` (bf16 MFMA operands, fp32 accumulate) is the throughput mode BASELINE
config 2 names: 2.8x the frames/s, sigmoid outputs within 7e-3, 1.4e-3 of the thresholded piano-roll cells differ (bench.py and
tests/test_gpu_model.py print both) -- opt in with `precision='bf16'`.  `'f16'` exists only in a library built with AMTX_BUILD_F16=1.
"""

import ctypes as C

import numpy as np
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, tools

__all__ = ['TranscriptionModel', 'OutputLayer', 'LogisticBank', 'AcousticModel', 'LanguageModel', 'OnsetsFrames',
           'SpectralFrontend']


class TranscriptionModel(nn.Module):
    """Generic transcription model (amt_tools/models/common.py:19-199)."""

    def __init__(self, dim_in, profile, in_channels=1, model_complexity=1, frame_width=1, device='cpu'):
        nn.Module.__init__(self)
        self.dim_in = dim_in
        self.profile = profile
        self.in_channels = in_channels
        self.model_complexity = model_complexity
        self.frame_width = frame_width
        self.device = device
        self.iter = 0
        # hook for on-device front-ends (models/common.py:56-57); empty by default
        self.frontend = nn.Sequential()

    def change_device(self, device=None):
        if device is None:
            device = self.device
        if isinstance(device, int):
            device = torch.device(f'cuda:{device}' if torch.cuda.is_available() else 'cpu')
        self.device = device
        self.to(self.device)
        if torch.device(self.device).type == 'cuda':
            # the training path's HIP convolutions / BatchNorm passes (amt_tools_amd/autograd.py) work on channels-last maps; keeping the
            # parameters in that memory format too means no layout copies around them.  Values, shapes and state_dict are unchanged
            self.to(memory_format=torch.channels_last)
        # an on-device front-end follows the model (its plans / tables are created per device on first use)
        for m in self.frontend:
            mod = getattr(m, 'module', None)
            if mod is not None and hasattr(mod, 'change_device'):
                mod.change_device(self.device)

    def pre_proc(self, batch):
        """To-device copy (the caller's dict is not modified) + optional front-end on the raw audio
        (models/common.py:83-122)."""
        batch = tools.dict_to_device(batch, self.device)
        audio = tools.unpack_dict(batch, tools.KEY_AUDIO)
        feats = tools.unpack_dict(batch, tools.KEY_FEATS)
        if audio is not None and len(self.frontend):
            frontend_feats = self.frontend(audio.unsqueeze(-2))
            feats = frontend_feats if feats is None else torch.cat((feats, frontend_feats), dim=1)
        batch[tools.KEY_FEATS] = feats
        return batch

    def forward(self, feats):
        raise NotImplementedError

    def post_proc(self, batch):
        raise NotImplementedError

    def run_on_batch(self, batch):
        batch = self.pre_proc(batch)
        batch[tools.KEY_OUTPUT] = self(batch[tools.KEY_FEATS])
        output = self.post_proc(batch)
        if tools.query_dict(batch, tools.KEY_TIMES):
            output[tools.KEY_TIMES] = batch[tools.KEY_TIMES]
        return output

    @classmethod
    def model_name(cls):
        return cls.__name__


class OutputLayer(nn.Module):
    def __init__(self, dim_in, dim_out, weights=None):
        super().__init__()
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.weights = None
        if weights is not None:
            self.set_weights(np.asarray(weights).flatten())

    def set_weights(self, weights, device='cpu'):
        if isinstance(device, int):
            device = torch.device(f'cuda:{device}' if torch.cuda.is_available() else 'cpu')
        self.weights = torch.Tensor(weights).to(device)


class LogisticBank(OutputLayer):
    """Multi-label logistic output layer (amt_tools/models/common.py:486-620)."""

    def __init__(self, dim_in, dim_out, weights=None):
        super().__init__(dim_in, dim_out, weights)
        self.output_layer = nn.Linear(self.dim_in, self.dim_out)

    def forward(self, feats):
        if feats.is_cuda and torch.is_grad_enabled():
            from .autograd import linear, linear_supported, note_fallback     # training on a GPU: the HIP GEMMs, forward and backward
            if linear_supported(feats, self.output_layer.weight):
                return linear(feats, self.output_layer.weight, self.output_layer.bias)
            note_fallback('LogisticBank.output_layer', f'Linear {self.dim_in} -> {self.dim_out}, {feats.dtype}')
        return self.output_layer(feats)

    def get_loss(self, estimated, reference):
        """BCE with logits; mean over frames, sum over keys, mean over the batch (common.py:541-584).
        estimated (B,T,O) logits, reference (B,O,T)."""
        if (estimated.is_cuda and estimated.dtype == torch.float32 and estimated.dim() == 3 and reference.dim() == 3
                and reference.shape == (estimated.shape[0], estimated.shape[2], estimated.shape[1])):
            from .autograd import bce_logits_loss     # one HIP pass: loss + d loss / d logits
            return bce_logits_loss(estimated, reference, self.weights)
        if estimated.is_cuda and torch.is_grad_enabled() and estimated.requires_grad:
            # (a validation pass that computes a loss on logits of another dtype / layout is not a training step: nothing to record)
            from .autograd import note_fallback
            note_fallback('LogisticBank.get_loss', f'logits {tuple(estimated.shape)} {estimated.dtype} vs labels {tuple(reference.shape)}')
        est = estimated.transpose(-2, -1)
        weight = self.weights.unsqueeze(-1) if self.weights is not None else None
        loss = F.binary_cross_entropy_with_logits(est.float(), reference.float(), weight=weight, reduction='none')
        return loss.mean(dim=-1).sum(dim=-1).mean()

    def finalize_output(self, raw_output, threshold=None):
        """sigmoid -> (B,O,T) -> optional threshold (common.py:586-620)."""
        final = torch.sigmoid(raw_output.clone().detach()).transpose(-2, -1)
        if threshold is not None:
            final = tools.threshold_activations(final.contiguous(), threshold)
        return final


class AcousticModel(nn.Module):
    """Kelz-style acoustic model (amt_tools/models/onsetsframes.py:330-463); module layout kept for
    state_dict compatibility: layer{1,2,3}.0 = Conv2d, .1 = BatchNorm2d, fc1.0 = Linear."""

    def __init__(self, dim_in, dim_out, in_channels=1, model_complexity=2):
        super().__init__()
        nf1 = nf2 = 16 * model_complexity
        nf3 = 32 * model_complexity
        self.layer1 = nn.Sequential(nn.Conv2d(in_channels, nf1, (3, 3), padding=1), nn.BatchNorm2d(nf1), nn.ReLU())
        self.layer2 = nn.Sequential(nn.Conv2d(nf1, nf2, (3, 3), padding=1), nn.BatchNorm2d(nf2), nn.ReLU(),
                                    nn.MaxPool2d((1, 2)), nn.Dropout(0.25))
        self.layer3 = nn.Sequential(nn.Conv2d(nf2, nf3, (3, 3), padding=1), nn.BatchNorm2d(nf3), nn.ReLU(),
                                    nn.MaxPool2d((1, 2)), nn.Dropout(0.25))
        self.fc1 = nn.Sequential(nn.Linear(nf3 * (dim_in // 4), dim_out), nn.Dropout(0.50))

    use_hip_bn = True        # False: stock nn.BatchNorm2d / ReLU / MaxPool2d (ATen + MIOpen) also on the GPU in training mode

    def _stage(self, layer, x):
        """One conv stage.  Training on a GPU: the convolution is an implicit GEMM on the HIP kernels (autograd.conv3x3: forward,
        input and weight gradients), BatchNorm (batch statistics) + ReLU (+ MaxPool) run as the HIP passes of
        amt_tools_amd/autograd.py (bn_relu_pool); the Dropout behind them is the module's own."""
        mods = list(layer)
        if self.training and self.use_hip_bn and x.is_cuda and not isinstance(mods[1], nn.SyncBatchNorm):
            from .autograd import bn_relu_pool, bn_relu_pool_supported, conv3x3, conv3x3_supported, note_fallback
            if conv3x3_supported(x, mods[0]):
                y = conv3x3(x, mods[0])
            else:
                note_fallback('AcousticModel conv', f'Conv2d {mods[0].in_channels} -> {mods[0].out_channels} on {x.dtype}')
                y = mods[0](x)
            if bn_relu_pool_supported(y, mods[1]):
                pool = len(mods) > 3 and isinstance(mods[3], nn.MaxPool2d)
                y = bn_relu_pool(y, mods[1], pool)
                for m in mods[(4 if pool else 3):]:
                    y = m(y)
                return y
            note_fallback('AcousticModel BatchNorm+ReLU+MaxPool', f'{y.shape[1]} channels, {y.dtype}, channels_last={y.is_contiguous(memory_format=torch.channels_last)}')
            for m in mods[1:]:
                y = m(y)
            return y
        if self.training and x.is_cuda:
            from .autograd import note_fallback
            note_fallback('AcousticModel stage', 'use_hip_bn off' if not self.use_hip_bn else 'SyncBatchNorm (stock modules by design: its statistics are a collective)')
        return layer(x)

    def forward(self, in_feats):
        x = self._stage(self.layer3, self._stage(self.layer2, self._stage(self.layer1, in_feats)))
        if (self.training and x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                and not x.is_contiguous()):
            # channels-last map: (B,T,F,C) is a free view, so the reference's (channel, freq) flatten order is applied to fc1's
            # weight columns (7.5 MB, differentiable permute) instead of transposing the activations (73 MB per head and step,
            # forward and backward)
            lin = self.fc1[0]
            B, C, T, F = x.shape
            w = lin.weight.view(lin.out_features, C, F).transpose(1, 2).reshape(lin.out_features, F * C)
            from .autograd import linear, linear_supported, note_fallback
            xin = x.permute(0, 2, 3, 1).reshape(B, T, F * C)
            if linear_supported(xin, w):
                y = linear(xin, w, lin.bias)
            else:
                note_fallback('AcousticModel.fc1', f'Linear {w.shape[1]} -> {w.shape[0]}, {xin.dtype}')
                y = torch.nn.functional.linear(xin, w, lin.bias)
            for m in list(self.fc1)[1:]:
                y = m(y)
            return y
        if self.training and x.is_cuda:
            from .autograd import note_fallback
            note_fallback('AcousticModel.fc1', 'conv map not in channels-last layout')
        x = x.transpose(-3, -2).flatten(-2)
        return self.fc1(x)


class LanguageModel(nn.Module):
    """BiLSTM language model (amt_tools/models/onsetsframes.py:466-575).  The reference processes eval
    inputs in `chunk_len`-frame chunks carrying the state; that is numerically the full-sequence
    result (SURVEY finding F9), which is what this module computes."""

    def __init__(self, dim_in, dim_out, chunk_len=512, bidirectional=True):
        super().__init__()
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.chunk_len = chunk_len
        self.num_directions = int(bidirectional) + 1
        self.hidden_size = self.dim_out // self.num_directions
        self.mlm = nn.LSTM(input_size=self.dim_in, hidden_size=self.hidden_size, batch_first=True, bidirectional=bidirectional)
        self.use_hip_autograd = True      # False: stock nn.LSTM (ATen / MIOpen) also on the GPU

    def forward(self, in_feats):
        # differentiable path on a GPU (training, or eval outside the engine): both directions' recurrence is one persistent HIP
        # kernel forward and one backward (amt_tools_amd/autograd.py) instead of MIOpen's per-time-step LSTM
        from .autograd import HIDDEN_SIZES
        if (in_feats.is_cuda and in_feats.dtype == torch.float32 and self.num_directions == 2 and self.hidden_size in HIDDEN_SIZES
                and self.mlm.num_layers == 1 and self.use_hip_autograd):
            from .autograd import bilstm
            return bilstm(in_feats, self.mlm)
        if in_feats.is_cuda and torch.is_grad_enabled():
            from .autograd import note_fallback
            note_fallback('LanguageModel', f'nn.LSTM hidden {self.hidden_size} x {self.num_directions} directions, {in_feats.dtype}, '
                                           f'use_hip_autograd={self.use_hip_autograd}')
        return self.mlm(in_feats)[0]


class SpectralFrontend(nn.Module):
    """`TranscriptionModel.frontend` entry wrapping a GPU FeatureModule: (B,1,N) audio -> (B,C,F,T).
    The tensor is produced in the model's (T,F)-major memory order and returned as a transposed view, so
    the model's own transpose (onsetsframes.py:90) yields contiguous data without a copy."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, audio):
        assert audio.dim() == 3 and audio.shape[1] == 1
        from .features import _SpecPlanOwner
        if isinstance(self.module, _SpecPlanOwner):
            feats = self.module.process_batch(audio[:, 0].float(), model_layout=True)    # (B,1,T,F)
            return feats.transpose(-1, -2)
        return self.module.process_batch(audio[:, 0].float())                            # CQT family: (B,C,F,T)


def _pick_side_stream(device, candidates=6, spins=20, spin_cycles=20000):
    """A side stream that really runs BESIDE the caller's current stream, for the training step's pitch head.  HIP multiplexes a process's
    streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues in creation order, and two streams that land on the same queue execute one
    after the other.  Measured in round 6: as soon as the process holds an RCCL communicator -- whose own streams shift the order -- the
    first `torch.cuda.Stream()` sits on the null stream's queue, and the training step ran its two heads back to back: 11.4 instead of 9.5
    ms (GPU_MAX_HW_QUEUES=8 / 16: 9.5; =1: 11.3; the data-parallel step of BASELINE config 4 is exactly that case).  So the stream is chosen
    by a test: a chain of short one-block spin kernels (torch.cuda._sleep) on the current stream and on the candidate at once takes ~1.4 x
    one chain's time when the two overlap and ~1.9 x when they share a queue (tools/stream_queue_probe.py: the same verdicts as a chain of 40
    small GEMMs per stream; ONE long spin kernel is not a reliable probe against the null stream).  The first candidate below 1.7 x is
    kept; without one (or without _sleep) the first stream created.  Running BOTH heads on tested streams of their own costs 0.45 ms of
    extra cross-stream waits per step (measured 9.9 vs 9.45 ms), so the recurrent heads stay on the caller's stream."""
    import time
    sleep = getattr(torch.cuda, '_sleep', None)
    main = torch.cuda.current_stream(device)
    first = None
    try:
        with torch.cuda.device(device):
            def chains(streams):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for st in streams:
                    with torch.cuda.stream(st):
                        for _ in range(spins):
                            sleep(spin_cycles)
                for st in streams:
                    st.synchronize()
                return time.perf_counter() - t0
            for _ in range(candidates):
                s = torch.cuda.Stream(device=device)
                first = first or s
                if sleep is None:
                    return s
                chains((main, s))                               # first use of the stream (and of the spin kernel)
                one = min(chains((main,)) for _ in range(2))
                if min(chains((main, s)) for _ in range(2)) < 1.7 * one:
                    return s
    except Exception:                                           # noqa: BLE001 -- a probe: any stream is a correct stream
        pass
    return first or torch.cuda.Stream(device=device)


class PendingFeatures(object):
    """What OnsetsFrames.pre_proc puts under KEY_FEATS inside run_on_batch when the dB scaling of the front-end is deferred into the
    engine's first conv kernel (amtx_of_forward_power): the raw power mel spectrogram (B,T,F), the clips' own maxima (B,) and the
    FeatureModule that would finish them.  `materialize()` is the ordinary feature tensor (B,1,T,F), bit-identical to what the conv
    kernel stages."""

    def __init__(self, module, power, clip_max, ref=None):
        # ref: optional (B,) reference powers for the dB scale (track-level reference); None = each clip's own maximum
        self.module, self.power, self.clip_max, self.ref = module, power, clip_max, ref

    @property
    def device(self):
        return self.power.device

    def materialize(self):
        return self.module.scale_batch(self.power, self.clip_max, self.ref, model_layout=True)


class PendingFeatures16(PendingFeatures):
    """What OnsetsFrames.pre_proc puts under KEY_FEATS inside run_on_batch when a CQT-family front-end has written its features in the
    first conv kernel's own staging format (FeatureModule.process_batch16 -> amtx_of_forward_feats16): (B,T,F,8) bfloat16, the harmonics
    of a position side by side.  `materialize()` is the ordinary fp32 feature tensor as a (B,C,T,F) view, computed from the audio."""

    def __init__(self, module, feats16, audio):
        self.module, self.feats16, self.audio = module, feats16, audio

    @property
    def device(self):
        return self.feats16.device

    def materialize(self):
        return self.module.process_batch(self.audio).transpose(-1, -2)


class _OFEngine(object):
    """ctypes handle of an amtx_of_model + its workspace, bound to one device."""

    def __init__(self, model, device):
        self.device = device
        self.handle = C.c_void_p()
        prec = {'bf16': 0, 'x3': 1, 'f16': 2}[model.precision]
        L = _lib.lib()
        with torch.cuda.device(device):
            _lib.check(L.amtx_of_model_create(C.byref(self.handle), int(model.dim_in), int(model.in_channels),
                                              int(model.model_complexity), int(model.profile.get_range_len()),
                                              int(model.has_offsets), prec), 'amtx_of_model_create')
        self.version = None
        self.workspace = None
        self.device_sync = os.environ.get('AMTX_HOST_WEIGHT_SYNC') is None     # A/B switch: always pack on the host
        self.device_syncs = 0                            # re-syncs that stayed on the GPU (tests)

    def sync_weights(self, model):
        sd = model.state_dict()
        version = tuple((k, v._version, v.data_ptr()) for k, v in sd.items() if v.dtype.is_floating_point)
        if version == self.version:
            return
        L = _lib.lib()
        items = [(k, v) for k, v in sd.items() if v.dtype.is_floating_point and not k.startswith('frontend.') and v.numel() > 0]
        # A RE-sync (validate() inside train() pays one at every checkpoint) stays on the GPU where the library can pack there: the
        # tensors are handed over as device pointers and folded / packed by kernels -- the same bits as the host path.
        if self.version is not None and self.device_sync and all(v.is_cuda and v.device == torch.device(self.device) and v.dtype == torch.float32 for _, v in items):
            keep = []                                   # channels-last convolution weights (the GPU training layout) go through a dense device copy
            for k, v in items:
                t = v.detach()
                t = t if t.is_contiguous() else t.contiguous()
                keep.append(t)
                _lib.check(L.amtx_of_model_set_tensor_device(self.handle, k.encode(), _lib.ptr(t), t.numel()), 'amtx_of_model_set_tensor_device')
            with torch.cuda.device(self.device):
                rc = L.amtx_of_model_finalize_device(self.handle, _lib.current_stream(self.device))
                if keep and rc == 0:
                    torch.cuda.current_stream(self.device).synchronize()     # the borrowed copies may go once the pack kernels have read them
            del keep
            if rc == 0:
                self.version = version
                self.device_syncs += 1
                return
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, 'amtx_of_model_finalize_device')
            self.device_sync = False                    # this configuration packs on the host
        # ONE device-to-host copy of all parameters and buffers (a copy per tensor is ~60 synchronisations per re-sync); the library packs
        # from host memory
        flat = torch.cat([v.detach().reshape(-1).to(torch.float32) for _, v in items]).cpu().numpy()
        off = 0
        for k, v in items:
            n = v.numel()
            arr = flat[off:off + n]
            off += n
            _lib.check(L.amtx_of_model_set_tensor(self.handle, k.encode(), _lib.ptr(arr), n), 'amtx_of_model_set_tensor')
        with torch.cuda.device(self.device):
            _lib.check(L.amtx_of_model_finalize(self.handle), 'amtx_of_model_finalize')
        self.version = version

    def fuses_db_scale(self):
        return bool(_lib.lib().amtx_of_fuses_db_scale(self.handle))

    def takes_feats16(self):
        """0: no; 1: (B,T,F,8) 16-bit channels-last features (one-plane engine); 2: the same as two planes (2,B,T,F,8) (x3 engine, round 6)."""
        return int(_lib.lib().amtx_of_takes_feats16(self.handle))

    def conv_stack_fused(self, batch, num_frames):
        """True when a forward pass of this shape runs the three convolution layers as one kernel (csrc/convf.hip)."""
        return bool(_lib.lib().amtx_of_conv_stack_fused(self.handle, int(batch), int(num_frames)))

    def forward(self, feats, want_logits=True):
        """feats: (B,C,T,F) fp32 CUDA tensor (any strides), or PendingFeatures.  Returns binary maps + raw logits."""
        L = _lib.lib()
        pending = feats if isinstance(feats, PendingFeatures) else None
        pending16 = isinstance(pending, PendingFeatures16)
        if pending16:
            feats = pending.feats16                         # (B,T,F,8) or two planes (2,B,T,F,8): shape and device only
            B, T = feats.shape[-4], feats.shape[-3]
        elif pending is not None:
            feats = pending.power.unsqueeze(1)
        if not pending16:
            B, Cc, T, Fd = feats.shape
        need = L.amtx_of_workspace_bytes(self.handle, B, T)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = _lib.alloc_workspace(need, feats.device)
        n_out = self.n_out
        opts = dict(dtype=torch.float32, device=feats.device)
        onsets = torch.empty((B, n_out, T), **opts)
        multi_pitch = torch.empty((B, n_out, T), **opts)
        lo = torch.empty((B, T, n_out), **opts) if want_logits else None
        lm = torch.empty((B, T, n_out), **opts) if want_logits else None
        lp = torch.empty((B, T, n_out), **opts) if want_logits else None
        sb, sc, st, sf = (0, 0, 0, 0) if pending16 else feats.stride()
        with torch.cuda.device(feats.device):
            if pending16:
                _lib.check(L.amtx_of_forward_feats16(self.handle, _lib.ptr(feats), B, T, _lib.ptr(self.workspace), self.workspace.numel(),
                                                     _lib.ptr(onsets), _lib.ptr(multi_pitch), _lib.ptr(lo), _lib.ptr(lm), _lib.ptr(lp),
                                                     _lib.current_stream(feats.device)), 'amtx_of_forward_feats16')
            elif pending is not None:
                _lib.check(L.amtx_of_forward_power(self.handle, _lib.ptr(feats), sb, st, sf, _lib.ptr(pending.clip_max), _lib.ptr(pending.ref), B, T,
                                                   _lib.ptr(self.workspace), self.workspace.numel(), _lib.ptr(onsets), _lib.ptr(multi_pitch),
                                                   _lib.ptr(lo), _lib.ptr(lm), _lib.ptr(lp), _lib.current_stream(feats.device)),
                           'amtx_of_forward_power')
            else:
                _lib.check(L.amtx_of_forward(self.handle, _lib.ptr(feats), sb, sc, st, sf, B, T, _lib.ptr(self.workspace),
                                             self.workspace.numel(), _lib.ptr(onsets), _lib.ptr(multi_pitch), _lib.ptr(lo),
                                             _lib.ptr(lm), _lib.ptr(lp), _lib.current_stream(feats.device)), 'amtx_of_forward')
        self._last = (B, T)
        return onsets, multi_pitch, lo, lm, lp

    def offsets(self, device, want_logits=True):
        """OnsetsFrames2: offset probabilities (B,O,T) [+ raw logits (B,T,O)] of the last forward."""
        L = _lib.lib()
        B, T = self._last
        opts = dict(dtype=torch.float32, device=device)
        prob = torch.empty((B, self.n_out, T), **opts)
        logits = torch.empty((B, T, self.n_out), **opts) if want_logits else None
        with torch.cuda.device(device):
            _lib.check(L.amtx_of_offsets(self.handle, _lib.ptr(self.workspace), self.workspace.numel(), B, T, _lib.ptr(prob),
                                         _lib.ptr(logits), _lib.current_stream(device)), 'amtx_of_offsets')
        return prob, logits

    def __del__(self):
        try:
            _lib.lib().amtx_of_model_destroy(self.handle)
        except Exception:
            pass


class OnsetsFrames(TranscriptionModel):
    """Onsets & Frames V1 (amt_tools/models/onsetsframes.py:17-196)."""

    has_offsets = False

    def __init__(self, dim_in, profile, in_channels=1, model_complexity=2, detach_heads=False, device='cpu',
                 precision='x3'):
        super().__init__(dim_in, profile, in_channels, model_complexity, 1, device)
        assert precision in ('bf16', 'x3', 'f16')
        self.detach_heads = detach_heads
        self.precision = precision
        self.dim_am = 256 * self.model_complexity
        self.dim_lm = 256 * (self.model_complexity - 1)
        dim_out = self.profile.get_range_len()
        self.onset_head = nn.Sequential(AcousticModel(self.dim_in, self.dim_am, self.in_channels, self.model_complexity),
                                        LanguageModel(self.dim_am, self.dim_lm), LogisticBank(self.dim_lm, dim_out))
        self.pitch_head = nn.Sequential(AcousticModel(self.dim_in, self.dim_am, self.in_channels, self.model_complexity),
                                        LogisticBank(self.dim_am, dim_out))
        self.dim_aj = 2 * dim_out
        self.adjoin = nn.Sequential(LanguageModel(self.dim_aj, self.dim_lm), LogisticBank(self.dim_lm, dim_out))

    # ---- engine management (device handles never enter state_dict / pickles, SURVEY finding F11) ----
    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_engine', None)
        state.pop('_engine_out', None)
        state.pop('_engine_offsets', None)
        state.pop('_side_stream', None)
        state.pop('_overlap_armed', None)
        state.pop('_fb_last_forward', None)
        state.pop('_overlap_latched_off', None)
        return state

    def _get_engine(self, device):
        eng = self.__dict__.get('_engine')
        if eng is None or eng.device != device:
            eng = _OFEngine(self, device)
            eng.n_out = self.profile.get_range_len()
            self.__dict__['_engine'] = eng
        eng.sync_weights(self)
        return eng

    def pre_proc(self, batch):
        pending = self._deferred_scale(batch) if self.__dict__.get('_in_run_on_batch') else None
        if pending is not None:
            return pending
        batch = super().pre_proc(batch)
        # frequency <-> time (onsetsframes.py:90); a view, the engine takes strides
        batch[tools.KEY_FEATS] = batch[tools.KEY_FEATS].transpose(-1, -2)
        return batch

    def _deferred_scale(self, batch):
        """Inside run_on_batch, eval mode, raw audio only, a front-end of this package on the GPU and an engine whose first conv stages the
        features itself.  dB-scaled log-mel / STFT: the front-end stops after its power kernel and the dB scaling happens inside the conv
        kernel (PendingFeatures; one kernel launch and one write + read of the feature tensor less per batch).  CQT family (HCQT / HVQT: one
        channel per harmonic): the front-end writes its map in the conv kernel's own staging format (PendingFeatures16: one 16-bit plane for
        the bf16 engine, the two planes of the split for x3).
        Returns the pre-processed batch, or None when none of that holds (the ordinary path then runs)."""
        if self.training or len(self.frontend) != 1 or not isinstance(self.frontend[0], SpectralFrontend):
            return None
        if os.environ.get('AMTX_DEFER_DB_SCALE', '1') == '0':     # A/B switch: front-end writes finished features (amtx_spec_scale)
            return None
        from .features import _SpecPlanOwner, _CqtPlanOwner
        module = self.frontend[0].module
        cqt = isinstance(module, _CqtPlanOwner)
        if cqt:
            # CQT family (HCQT: one channel per harmonic): the front-end writes the features in the conv kernel's staging format
            if os.environ.get('AMTX_CQT_FEATS16', '1') == '0':     # A/B switch: fp32 (B,C,F,T) features, converted by the conv kernel
                return None
        elif not isinstance(module, _SpecPlanOwner) or not getattr(module, 'decibels', False):
            return None
        if tools.query_dict(batch, tools.KEY_FEATS) or not tools.query_dict(batch, tools.KEY_AUDIO):
            return None
        if torch.device(self.device).type != 'cuda':
            return None
        batch = tools.dict_to_device(batch, self.device)
        audio = batch[tools.KEY_AUDIO]
        if not (torch.is_tensor(audio) and audio.is_cuda and audio.dim() == 2):
            return None
        if cqt:
            form = self._get_engine(audio.device).takes_feats16()
            if not form:
                return None
            # amtx_of_forward_feats16 takes a bare (B,T,F,8) pointer and indexes it with the MODEL's dim_in / in_channels: a front-end of
            # another shape goes through the ordinary path, whose strides and shapes are explicit (and which raises on a mismatch)
            if int(getattr(module, 'n_bins', -1)) != int(self.dim_in) or int(module.get_num_channels()) != int(self.in_channels):
                return None
            audio = audio.float()
            batch[tools.KEY_FEATS] = PendingFeatures16(module, module.process_batch16(audio, split=(form == 2)), audio)
            return batch
        if not self._get_engine(audio.device).fuses_db_scale():
            return None
        power, clip_max = module.power_batch(audio.float())
        batch[tools.KEY_FEATS] = PendingFeatures(module, power, clip_max)
        return batch

    def run_on_batch(self, batch):
        # Without ground truth nothing downstream reads the raw logits (post_proc only thresholds them,
        # onsetsframes.py:186-194, and the engine has done that on the device): they stay in the engine's workspace
        # instead of being copied out into three (B,T,O) tensors.  forward() called on its own always returns logits.
        labelled = any(tools.query_dict(batch, k) for k in (tools.KEY_MULTIPITCH, tools.KEY_ONSETS, tools.KEY_OFFSETS))
        self.__dict__['_logits_wanted'] = labelled
        self.__dict__['_in_run_on_batch'] = True
        try:
            return super().run_on_batch(batch)
        finally:
            self.__dict__.pop('_logits_wanted', None)
            self.__dict__.pop('_in_run_on_batch', None)

    def forward(self, feats):
        """feats (B,C,T,F) -> dict of raw logits (B,T,O) under 'onsets' and 'multi_pitch'."""
        pending = isinstance(feats, PendingFeatures)
        if pending and self.training:
            feats, pending = feats.materialize(), False
        if pending or (feats.is_cuda and not self.training):
            eng = self._get_engine(feats.device)
            if not pending and feats.dtype != torch.float32:
                feats = feats.float()
            want = self.__dict__.get('_logits_wanted', True)
            onsets_bin, mp_bin, lo, lm, lp = eng.forward(feats if pending else feats.detach(), want_logits=want)
            if not want:
                # label-free run_on_batch: the entries are the final piano rolls already (post_proc passes them through)
                lo, lm = onsets_bin, mp_bin
            output = {tools.KEY_ONSETS: lo, tools.KEY_MULTIPITCH: lm}
            # piano rolls already thresholded on the device; post_proc picks them up for these logits
            self.__dict__['_engine_out'] = (lo, lm, onsets_bin, mp_bin, lp)
            if self.has_offsets:
                prob, logits = eng.offsets(feats.device, want_logits=want)
                output[tools.KEY_OFFSETS] = logits if want else prob
                self.__dict__['_engine_offsets'] = (output[tools.KEY_OFFSETS], prob)
            return output
        self.__dict__.pop('_engine_out', None)
        self.__dict__.pop('_engine_offsets', None)
        output = dict()
        fb0 = None
        if feats.is_cuda and self.training and torch.is_grad_enabled():
            from . import autograd as _ag
            fb0 = _ag.fallback_total()
        if self._overlap_heads(feats):
            # The detector heads only meet at the refinement stage (onsetsframes.py:118-134): the pitch head runs on a side stream beside the
            # recurrent heads -- acoustic model + a 625-step BiLSTM that keeps 4 of the 256 CUs busy at 8 clips; autograd replays
            # every op's backward on the stream of its forward, so the backward passes overlap the same way (11.0 -> 9.85 ms per step).
            # Tensors that cross streams are handed to the allocator with record_stream.  History: round 1 saw intermittent hangs with two
            # streams while the convolutions / Linear layers still ran on MIOpen / hipBLASLt; with the all-HIP step tools/two_stream_repro.py
            # and tests/test_gpu_rccl.py run thousands of steps clean (HISTORY.md "Measured (round 4)").  AMTX_TRAIN_OVERLAP=0 keeps one stream.
            main = torch.cuda.current_stream(feats.device)
            side = self.__dict__.get('_side_stream')
            if side is None or side.device != feats.device:
                side = _pick_side_stream(feats.device)       # tested to run beside the caller's stream
                self.__dict__['_side_stream'] = side
            side.wait_stream(main)
            feats.record_stream(side)
            with torch.cuda.stream(side):          # issued first, as in the reference (onsetsframes.py:121-124): same Dropout RNG order
                multi_pitch = self.pitch_head(feats)
            onsets, offsets = self._recurrent_heads(feats)
            main.wait_stream(side)
            multi_pitch.record_stream(main)
        else:
            multi_pitch = self.pitch_head(feats)
            onsets, offsets = self._recurrent_heads(feats)
        output[tools.KEY_ONSETS] = onsets
        heads = [onsets]
        if self.has_offsets:
            output[tools.KEY_OFFSETS] = offsets
            heads.append(offsets)
        if self.detach_heads:
            heads = [h.clone().detach() for h in heads]
        joint = torch.cat(heads + [multi_pitch], -1)
        output[tools.KEY_MULTIPITCH] = self.adjoin(joint)
        if fb0 is not None:
            self.__dict__['_fb_last_forward'] = _ag.fallback_total() - fb0     # 0: the next training forward may overlap its heads
            if self.__dict__['_fb_last_forward']:
                self.__dict__['_overlap_latched_off'] = True                   # sticky: see _overlap_heads
        return output

    def _overlap_heads(self, feats):
        """Training on a GPU: pitch head on a side stream beside the recurrent heads, unless AMTX_TRAIN_OVERLAP=0 (or the attribute
        `overlap_heads = False` on the model) -- and only while every dense layer of the step runs on this package's HIP kernels.
        tools/two_stream_repro.py narrowed the round-1 hang down to the vendor libraries: with BOTH MIOpen convolutions and hipBLASLt
        GEMMs in the two heads the side stream stops in the first head's fc1 GEMM within 2 - 90 steps (either library alone, or the
        all-HIP step: thousands of steps clean).  So the first training forward of a model runs on one stream, and the overlap is armed
        only if THIS model's previous training forward recorded no ATen fallback (bracketed with autograd.fallback_total() in forward():
        fallbacks of other models or of a validation pass do not count).  The side stream is picked by a concurrency test
        (_pick_side_stream): a stream that shares a hardware queue with the caller's does not overlap anything."""
        if not (feats.is_cuda and self.training and torch.is_grad_enabled()):
            return False
        if os.environ.get('AMTX_TRAIN_OVERLAP', '1') == '0' or not self.__dict__.get('overlap_heads', True):
            return False
        from . import autograd as ag
        if not ag.USE_HIP_DENSE:
            return False
        # Several ranks (round 6): on by default too.  Rounds 4 - 5 kept it off under a real process group "until an 8-GPU soak has run"; what
        # is known since: the round-1 hang needs MIOpen + hipBLASLt kernels on the two streams (none in this step); 300 steps with a one-rank
        # RCCL communicator and both streams run clean and bit-identical (tests/test_gpu_rccl.py); two rank processes with both streams each
        # run on one GPU (tests/test_gpu_multirank.py); and the gradient all-reduce is enqueued only after autograd has joined the side stream
        # back into the main one and the next forward forks only after the optimizer step -- RCCL's kernels never run beside the two-stream
        # phase.  AMTX_TRAIN_OVERLAP=0 is the switch back.
        # THIS model's previous training forward must have run without an ATen fallback (the count is bracketed per forward in forward():
        # other models, or a validation pass, do not switch the overlap off for the rest of the process).  The switch is STICKY per model:
        # one training forward of this model that recorded a fallback (an odd-shaped batch sent a layer to MIOpen / hipBLASLt) latches the
        # overlap off for the life of the object -- a vendor-library kernel under two streams is the hang tools/two_stream_repro.py found,
        # and a batch shape that fell back once can come by again at any step.  The bracket covers forward() only: LogisticBank.get_loss in
        # post_proc is not in it, and needs not be (its ATen branch is elementwise BCE on the main stream, after the streams have joined).
        if self.__dict__.get('_overlap_latched_off', False):
            return False
        if self.__dict__.get('_fb_last_forward', None) != 0:
            return False
        return True

    def _recurrent_heads(self, feats):
        """Onset (and offset) detector heads: (onsets, offsets or None)."""
        if self._grouped_recurrences(feats):
            # the onset and offset heads are independent: their two recurrences run as ONE grouped launch each way
            from .autograd import bilstm_multi
            (am_on, lm_on, lb_on), (am_off, lm_off, lb_off) = self.onset_head, self.offset_head
            l_on, l_off = bilstm_multi([am_on(feats), am_off(feats)], [lm_on.mlm, lm_off.mlm])
            return lb_on(l_on), lb_off(l_off)
        onsets = self.onset_head(feats)
        offsets = self.offset_head(feats) if self.has_offsets else None
        return onsets, offsets

    def _grouped_recurrences(self, feats):
        """Training on a GPU with an offset head whose LSTM and the onset head's are both on the HIP autograd path."""
        if not (self.has_offsets and feats.is_cuda and self.training and feats.dtype == torch.float32):
            return False
        from .autograd import HIDDEN_SIZES
        lms = (self.onset_head[1], self.offset_head[1])
        return all(isinstance(lm, LanguageModel) and lm.use_hip_autograd and lm.num_directions == 2 and lm.mlm.num_layers == 1
                   and lm.hidden_size in HIDDEN_SIZES for lm in lms) and lms[0].hidden_size == lms[1].hidden_size

    def post_proc(self, batch):
        """Losses (when ground truth is present) + final piano rolls (onsetsframes.py:138-196)."""
        output = batch[tools.KEY_OUTPUT]
        onset_layer, pitch_layer = self.onset_head[-1], self.adjoin[-1]
        onsets_est, multi_pitch_est = output[tools.KEY_ONSETS], output[tools.KEY_MULTIPITCH]
        if tools.KEY_MULTIPITCH in batch.keys():
            loss = dict()
            multi_pitch_ref = batch[tools.KEY_MULTIPITCH]
            loss[tools.KEY_LOSS_PITCH] = pitch_layer.get_loss(multi_pitch_est, multi_pitch_ref)
            if tools.KEY_ONSETS in batch.keys():
                onsets_ref = batch[tools.KEY_ONSETS]
            else:
                # intended behaviour of onsetsframes.py:176-178 (the reference's NumPy helper fails on tensors)
                onsets_ref = tools.multi_pitch_to_onsets(multi_pitch_ref)
            loss[tools.KEY_LOSS_ONSETS] = onset_layer.get_loss(onsets_est, onsets_ref)
            loss[tools.KEY_LOSS_TOTAL] = loss[tools.KEY_LOSS_PITCH] + loss[tools.KEY_LOSS_ONSETS]
            output[tools.KEY_LOSS] = loss
        cached = self.__dict__.pop('_engine_out', None)
        if cached is not None and cached[0] is onsets_est and cached[1] is multi_pitch_est:
            output[tools.KEY_ONSETS], output[tools.KEY_MULTIPITCH] = cached[2], cached[3]
        else:
            output[tools.KEY_ONSETS] = onset_layer.finalize_output(onsets_est, 0.5)
            output[tools.KEY_MULTIPITCH] = pitch_layer.finalize_output(multi_pitch_est, 0.5)
        return output

    def engine_logits(self, feats_bcft):
        """Raw logits of all three LogisticBanks from the HIP engine for features (B,C,F,T) on the GPU:
        dict with 'onsets', 'multi_pitch', 'pitch_head' (B,T,O) -- used by the parity tests."""
        feats = feats_bcft.transpose(-1, -2)
        eng = self._get_engine(feats.device)
        onsets_bin, mp_bin, lo, lm, lp = eng.forward(feats.float())
        return {'onsets': lo, 'multi_pitch': lm, 'pitch_head': lp, 'onsets_bin': onsets_bin, 'multi_pitch_bin': mp_bin}


class OnsetsFrames2(OnsetsFrames):
    """Onsets & Frames V2 (amt_tools/models/onsetsframes.py:199-327): adds an offset detector head whose logits join the
    refinement stage; `offsets` are returned as probabilities.  The HIP engine runs it at the model complexities
    `amtx_of_model_create` is built for (2, and 3 = the reference's default for this class); anything else raises there in eval
    mode on a GPU -- it never falls back."""

    has_offsets = True

    def __init__(self, dim_in, profile, in_channels=1, model_complexity=3, detach_heads=True, device='cpu', precision='x3'):
        super().__init__(dim_in, profile, in_channels, model_complexity, detach_heads, device, precision)
        dim_out = self.profile.get_range_len()
        self.offset_head = nn.Sequential(AcousticModel(self.dim_in, self.dim_am, self.in_channels, self.model_complexity),
                                         LanguageModel(self.dim_am, self.dim_lm), LogisticBank(self.dim_lm, dim_out))
        self.dim_aj += dim_out
        self.adjoin[0] = LanguageModel(self.dim_aj, self.dim_lm)

    def post_proc(self, batch):
        output = super().post_proc(batch)
        offset_layer = self.offset_head[-1]
        offsets_est = output[tools.KEY_OFFSETS]
        if tools.KEY_LOSS in output.keys():
            if tools.KEY_OFFSETS in batch.keys():
                offsets_ref = batch[tools.KEY_OFFSETS]
            else:
                offsets_ref = tools.multi_pitch_to_offsets(batch[tools.KEY_MULTIPITCH])
            loss = output[tools.KEY_LOSS]
            offsets_loss = offset_layer.get_loss(offsets_est, offsets_ref)
            loss[tools.KEY_LOSS_OFFSETS] = offsets_loss
            loss[tools.KEY_LOSS_TOTAL] += offsets_loss
            output[tools.KEY_LOSS] = loss
        cached = self.__dict__.pop('_engine_offsets', None)
        if cached is not None and cached[0] is offsets_est:
            output[tools.KEY_OFFSETS] = cached[1]
        else:
            output[tools.KEY_OFFSETS] = offset_layer.finalize_output(offsets_est)
        return output

    def engine_logits(self, feats_bcft):
        out = super().engine_logits(feats_bcft)
        prob, logits = self._get_engine(feats_bcft.device).offsets(feats_bcft.device)
        out['offsets'], out['offsets_prob'] = logits, prob
        return out


class SoftmaxGroups(OutputLayer):
    """One categorical distribution per degree of freedom: `num_groups` x `num_classes` logits per frame, the LAST class meaning
    "inactive" (labelled -1 in ground truth and in the final output).  Behaviour contract: amt_tools/models/common.py:305-483."""

    def __init__(self, dim_in, num_groups, num_classes, weights=None):
        self.num_groups = num_groups
        self.num_classes = num_classes
        super().__init__(dim_in, num_groups * num_classes, weights)
        self.output_layer = nn.Linear(self.dim_in, self.dim_out)

    def forward(self, feats):
        return self.output_layer(feats)

    def _grouped(self, logits):
        """(B, T, G*C) -> (B, T, G, C)"""
        return logits.reshape(logits.shape[0], -1, self.num_groups, self.num_classes)

    def get_loss(self, estimated, reference):
        """Negative log-likelihood of the labelled class of every group: summed over groups, averaged over frames, then over the
        batch; optional per-(group, class) weights multiply each term (what F.cross_entropy(weight=..., reduction='none') does).
        estimated (B, T, G*C) logits, reference (B, G, T) class indices with -1 for the inactive class."""
        log_prob = torch.log_softmax(self._grouped(estimated).float(), dim=-1)                 # (B, T, G, C)
        target = reference.transpose(-2, -1).long()                                             # (B, T, G)
        target = torch.where(target < 0, torch.full_like(target, self.num_classes - 1), target)
        nll = -log_prob.gather(-1, target.unsqueeze(-1)).squeeze(-1)                            # (B, T, G)
        if self.weights is not None:
            per_class = self.weights.to(nll.device).reshape(self.num_groups, self.num_classes)
            group_idx = torch.arange(self.num_groups, device=nll.device).expand_as(target)
            nll = nll * per_class[group_idx, target]
        return nll.sum(dim=-1).mean(dim=-1).mean()

    def finalize_output(self, raw_output, last_negative=True):
        """Most likely class per group and frame as (B, G, T) indices; the inactive class becomes -1 when `last_negative`."""
        winners = torch.softmax(self._grouped(raw_output.detach()), dim=-1).argmax(dim=-1)      # (B, T, G)
        if last_negative:
            winners = torch.where(winners == self.num_classes - 1, torch.full_like(winners, -1), winners)
        return winners.transpose(-2, -1)


class TabCNN(TranscriptionModel):
    """TabCNN, BASELINE config 1 (behaviour contract: amt_tools/models/tabcnn.py:17-221).  CPU plumbing on stock torch ops -- there
    is no HIP kernel work for this model; it exists so that the reference's CQT + TabCNN experiment runs
    against this package's FeatureModule / TranscriptionModel objects unchanged.  Module names and indices (`conv.{0,2,4}`,
    `dense.{0,3}`) are the reference's, so its checkpoints load; the 9-frame context windows are built on the model's device with a
    strided `unfold` view instead of the reference's device -> NumPy -> device round trip (tabcnn.py:122-127)."""

    CONTEXT = 9      # frames seen by one prediction (tabcnn.py:40)

    def __init__(self, dim_in, profile, in_channels=1, model_complexity=1, device='cpu'):
        super().__init__(dim_in, profile, in_channels, model_complexity, self.CONTEXT, device)
        self.online = False
        widths = (32 * model_complexity, 64 * model_complexity, 64 * model_complexity)
        layers, c_prev = [], self.in_channels
        for c in widths:
            layers += [nn.Conv2d(c_prev, c, (3, 3)), nn.ReLU()]
            c_prev = c
        self.conv = nn.Sequential(*layers, nn.MaxPool2d((2, 2)), nn.Dropout(0.25))
        # three unpadded 3x3 convolutions shave 6 off both axes, the pooling halves them
        self.conv_embedding_size = widths[-1] * ((self.dim_in - 6) // 2) * ((self.frame_width - 6) // 2)
        self.fc_embedding_size = 128 * model_complexity
        self.dense = nn.Sequential(nn.Linear(self.conv_embedding_size, self.fc_embedding_size), nn.ReLU(), nn.Dropout(0.50),
                                   SoftmaxGroups(self.fc_embedding_size, self.profile.get_num_dofs(), self.profile.num_pitches + 1))

    def toggle_online(self):
        self.online = not self.online

    def pre_proc(self, batch):
        """Features (B, C, F, T) -> one context window per frame, (B, T', C, F, W).  Offline: T' = T, the sequence is zero-padded
        by W // 2 frames on both sides; online: no padding (the caller supplies exactly the frames of a window)."""
        batch = super().pre_proc(batch)
        feats = batch[tools.KEY_FEATS]
        W = self.frame_width
        if not self.online:
            feats = F.pad(feats, (W // 2, W // 2))
        elif feats.shape[-1] < W:
            feats = F.pad(feats, ((W - feats.shape[-1]) // 2, W - feats.shape[-1] - (W - feats.shape[-1]) // 2))
        windows = feats.unfold(-1, W, 1)                                  # (B, C, F, T', W) view
        batch[tools.KEY_FEATS] = windows.permute(0, 3, 1, 2, 4)           # (B, T', C, F, W)
        return batch

    def forward(self, feats):
        B, T = feats.shape[:2]
        embeddings = self.conv(feats.reshape(B * T, self.in_channels, self.dim_in, self.frame_width))
        return {tools.KEY_TABLATURE: self.dense(embeddings.reshape(B, T, -1))}

    def post_proc(self, batch):
        output = batch[tools.KEY_OUTPUT]
        head = self.dense[-1]
        logits = output[tools.KEY_TABLATURE]
        if tools.KEY_TABLATURE in batch:
            output[tools.KEY_LOSS] = {tools.KEY_LOSS_TOTAL: head.get_loss(logits, batch[tools.KEY_TABLATURE])}
        output[tools.KEY_TABLATURE] = head.finalize_output(logits)
        return output
