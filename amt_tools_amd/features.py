"""
Host-side mirror of amt-tools' FeatureModule plugin API (amt_tools/features/*.py) for the hot path's
spectral front-end, backed by the HIP kernels behind include/amtx.h.

Same class names, constructor keywords, method set and return conventions as the reference
(FeatureModule: features/common.py:15-321; WaveformWrapper bookkeeping: features/waveform.py:43-185;
STFT: features/stft.py; MelSpec: features/mel.py), so a `TranscriptionDataset`, `FeatureStream` or the
paper scripts can be handed these objects unchanged:

    process_audio(audio: np.ndarray (N,)) -> np.ndarray (C, F, T) float32      # seam S1

plus two additions the reference does not have:

    process_batch(audio: torch.Tensor (B, N) on the GPU) -> torch.Tensor (B, C, F, T)   # batched, stays on device
    frontend()  -> nn.Module for `TranscriptionModel.frontend` (models/common.py:56-57,107-120): the
                   fused seam -- features are computed on the device inside `pre_proc`.

`librosa_version` selects the centre-padding convention of the librosa release the reference would
have resolved ('0.10': zeros, '0.9': reflect; SURVEY finding F12).

The GPU context is created lazily on first use (never at construction), so instances can be built
before DataLoader workers fork.  There is no CPU fallback: without the HIP extension or a GPU,
process_audio raises.
"""

import ctypes as C

import numpy as np

from . import _lib

__all__ = ['FeatureModule', 'WaveformWrapper', 'STFT', 'MelSpec', 'VQT', 'CQT', 'HVQT', 'HCQT']

FLOAT32 = 'float32'


class FeatureModule(object):
    """Generic feature-extraction module (features/common.py:15-321)."""

    def __init__(self, sample_rate, hop_length, num_channels, decibels=True):
        self.sample_rate = sample_rate
        self.hop_length = hop_length
        self.num_channels = num_channels
        self.decibels = decibels

    # ---- frame / sample bookkeeping (row A0) -------------------------------------------------
    def get_expected_frames(self, audio):
        num_frames = 0
        if audio.shape[-1] != 0:
            num_frames = 1 + len(audio) // self.hop_length
        return num_frames

    def get_sample_range(self, num_frames):
        sample_range = np.array([0])
        if num_frames > 0:
            max_samples = num_frames * self.hop_length - 1
            min_samples = max(1, max_samples - self.hop_length + 1)
            sample_range = np.arange(min_samples, max_samples + 1)
        return sample_range

    def get_num_samples_required(self):
        return self.get_sample_range(1)[-1]

    @staticmethod
    def divisor_pad(audio, divisor):
        pad_amt = divisor - (audio.shape[-1] % divisor)
        if pad_amt > 0 and pad_amt != divisor:
            audio = np.append(audio, np.zeros(pad_amt).astype(FLOAT32), axis=-1)
        return audio

    def frame_pad(self, audio):
        divisor = self.get_num_samples_required()
        if audio.shape[-1] > divisor:
            divisor = self.hop_length
        return self.divisor_pad(audio, divisor)

    def process_audio(self, audio):
        raise NotImplementedError

    # ---- host-array helpers kept for API parity (the device path fuses them into K2) --------
    def to_decibels(self, feats):
        """amplitude_to_db(ref=np.max) on a host array (features/common.py:181-201)."""
        mag = np.abs(np.asarray(feats))
        ref = mag.max() if mag.size else 1.0
        db = 10.0 * np.log10(np.maximum(1e-10, np.square(mag))) - 10.0 * np.log10(np.maximum(1e-10, ref ** 2))
        return np.maximum(db, db.max() - 80.0) if mag.size else db

    def post_proc(self, feats):
        if self.decibels:
            feats = self.to_decibels(feats)
            feats = feats / 80
            feats = feats + 1
        return np.expand_dims(feats, axis=0)

    def get_times(self, audio):
        frame_idcs = np.arange(self.get_expected_frames(audio))
        # librosa.frames_to_time: samples / sr in float64
        return (frame_idcs * self.hop_length).astype(np.float64) / float(self.sample_rate)

    def get_sample_rate(self):
        return self.sample_rate

    def get_hop_length(self):
        return self.hop_length

    def get_num_channels(self):
        return self.num_channels

    def get_feature_size(self):
        raise NotImplementedError

    @classmethod
    def features_name(cls):
        return cls.__name__


class WaveformWrapper(FeatureModule):
    """Frame bookkeeping of waveform-domain modules (features/waveform.py)."""

    def __init__(self, sample_rate=44100, hop_length=512, decibels=False, win_length=None, center=True):
        super().__init__(sample_rate=sample_rate, hop_length=hop_length, num_channels=1, decibels=decibels)
        if win_length is None:
            win_length = self.hop_length
        self.win_length = win_length
        self.center = center

    def get_expected_frames(self, audio):
        if self.center or audio.shape[-1] == 0:
            return super().get_expected_frames(audio)
        return 1 + ((max(0, (audio.shape[-1] - self.win_length)) - 1) // self.hop_length + 1)

    def get_sample_range(self, num_frames):
        if self.center or num_frames == 0:
            return super().get_sample_range(num_frames)
        if num_frames == 1:
            return np.arange(1, self.win_length + 1)
        return np.arange(1, self.hop_length + 1) + self.get_num_samples_required() + (num_frames - 2) * self.hop_length

    def center_pad(self, audio):
        half = int(self.win_length // 2)
        return np.pad(audio, [(half, half)], mode='constant')

    def process_audio(self, audio):
        """Raw framing (features/waveform.py:121-153): (win_length, T) strided frames, host only."""
        if audio.shape[-1] == 0:
            return np.zeros((self.win_length, 0))
        audio = self.center_pad(audio) if self.center else self.frame_pad(audio)
        n_frames = 1 + (audio.shape[-1] - self.win_length) // self.hop_length
        idx = np.arange(self.win_length)[:, None] + self.hop_length * np.arange(n_frames)[None, :]
        return audio[idx]

    def get_times(self, audio, at_start=False):
        times = super().get_times(audio)
        if self.center and at_start:
            times -= ((self.win_length // 2) / self.sample_rate)
        elif not self.center and not at_start:
            times += ((self.win_length // 2) / self.sample_rate)
        return times

    def get_feature_size(self):
        return self.win_length


def _index_of(device):
    """Device index of an int / str / torch.device ('cuda' without an index = the current device)."""
    import torch
    if isinstance(device, int):
        return device
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise _lib.AmtxError(f'the HIP front-end runs on a GPU, not on {dev}')
    return dev.index if dev.index is not None else torch.cuda.current_device()


class _SpecPlanOwner(object):
    """Lazily created amtx_spec_plan (device tables) shared by STFT / MelSpec instances."""

    def _plan_args(self):
        raise NotImplementedError

    def _get_plan(self, device=None):
        """The plan (device tables) for `device` (a torch.device / index; default: the module's own device).  Plans are cached per
        device index and created under that device, so a module built with the default device='cuda:0' can be handed tensors that
        live on another GPU (one process per GPU under torchrun: rank k's tensors are on cuda:k) without its kernels
        dereferencing cuda:0 allocations."""
        import torch
        index = self._device_index() if device is None else _index_of(device)
        plans = self.__dict__.setdefault('_plans', {})
        plan = plans.get(index)
        if plan is None:
            if not torch.cuda.is_available():
                raise _lib.AmtxError('no GPU visible: the spectral front-end has no CPU fallback')
            handle = C.c_void_p()
            with torch.cuda.device(index):
                _lib.check(_lib.lib().amtx_spec_plan_create(C.byref(handle), *self._plan_args()), 'amtx_spec_plan_create')
            plan = plans[index] = handle
        return plan

    def _device_index(self):
        return _index_of(self.device)

    def change_device(self, device):
        """Follow the model to its device (TranscriptionModel.change_device propagates here through SpectralFrontend)."""
        self.device = device

    def __getstate__(self):   # plans hold device pointers: never pickle them (DataLoader workers, torch.save)
        state = dict(self.__dict__)
        state.pop('_plans', None)
        state.pop('_prof_events', None)
        return state

    def __del__(self):
        for plan in self.__dict__.get('_plans', {}).values():
            try:
                _lib.lib().amtx_spec_plan_destroy(plan)
            except Exception:
                pass

    # ---- device path -------------------------------------------------------------------------
    def power_batch(self, audio):
        """K1 on a (B, N) float32 CUDA tensor -> (power (B,T,F) fp32, clip_max (B,))."""
        import torch
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
        audio = audio.contiguous()
        B, N = audio.shape
        plan = self._get_plan(audio.device)
        L = _lib.lib()
        T = _lib.check(L.amtx_spec_num_frames(plan, N), 'amtx_spec_num_frames')
        F = L.amtx_spec_num_bins(plan)
        power = torch.empty((B, T, F), dtype=torch.float32, device=audio.device)
        clip_max = torch.empty((B,), dtype=torch.float32, device=audio.device)
        with torch.cuda.device(audio.device):
            ev = self._prof_begin()
            _lib.check(L.amtx_spec_power(plan, _lib.ptr(audio), N, audio.stride(0) if B > 1 else N, B, _lib.ptr(power), _lib.ptr(clip_max),
                                         _lib.current_stream(audio.device)), 'amtx_spec_power')
            self._prof_end('spec_power', ev)
        return power, clip_max

    # optional per-kernel timing (bench.py): events are recorded on the current stream, which is the stream
    # the kernels are launched on
    def _prof_begin(self):
        if self.__dict__.get('_prof_events') is None:
            return None
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _prof_end(self, name, ev):
        if ev is None:
            return
        import torch
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self._prof_events.append((name, ev, end))

    def scale_batch(self, power, clip_max, ref=None, model_layout=False):
        """K2: dB / scale + layout.  Returns (B,1,F,T) (reference layout) or (B,1,T,F) (model layout)."""
        import torch
        B, T, F = power.shape
        L = _lib.lib()
        if self.decibels:
            transform = 0
        else:
            transform = 1 if self._linear_is_power else 2
        out = torch.empty((B, 1, T, F) if model_layout else (B, 1, F, T), dtype=torch.float32, device=power.device)
        with torch.cuda.device(power.device):
            ev = self._prof_begin()
            _lib.check(L.amtx_spec_scale(self._get_plan(power.device), _lib.ptr(power), _lib.ptr(clip_max), _lib.ptr(ref), B, T, transform,
                                         1 if model_layout else 0, _lib.ptr(out), _lib.current_stream(power.device)),
                       'amtx_spec_scale')
            self._prof_end('spec_scale', ev)
        return out

    def process_batch(self, audio, ref=None, model_layout=False):
        """Batched features on the device: (B, N) float32 CUDA tensor -> (B, 1, F, T) float32."""
        power, clip_max = self.power_batch(audio)
        return self.scale_batch(power, clip_max, ref, model_layout)

    def _process_host(self, audio, empty_rows):
        import torch
        if audio.shape[-1] == 0:
            return np.zeros((1, empty_rows, 0))
        dev = torch.device('cuda', self._device_index())
        x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(dev).unsqueeze(0)
        return self.process_batch(x)[0].cpu().numpy()

    def frontend(self):
        """nn.Module wrapper for `TranscriptionModel.frontend`: (B,1,N) audio -> (B,1,F,T) features."""
        from .models import SpectralFrontend
        return SpectralFrontend(self)


class STFT(WaveformWrapper, _SpecPlanOwner):
    """Magnitude spectrogram (features/stft.py) on the GPU."""
    _linear_is_power = False

    def __init__(self, sample_rate=16000, hop_length=512, decibels=True, win_length=None, center=True, n_fft=2048,
                 librosa_version='0.10', device='cuda:0'):
        self.n_fft = n_fft
        if win_length is None:
            win_length = self.n_fft
        WaveformWrapper.__init__(self, sample_rate=sample_rate, hop_length=hop_length, decibels=decibels,
                                 win_length=win_length, center=center)
        self.librosa_version = str(librosa_version)
        self.device = device

    def _pad_mode(self):
        return 1 if self.librosa_version.startswith('0.9') else 0

    def _plan_args(self):
        return (int(self.sample_rate), int(self.n_fft), int(self.hop_length), int(self.win_length), 0, 0,
                int(bool(self.center)), self._pad_mode())

    def process_audio(self, audio):
        # the reference reports n_fft rows for empty audio (features/stft.py:57-59)
        return self._process_host(audio, self.n_fft)

    def get_feature_size(self):
        return self.n_fft // 2 + 1


class MelSpec(STFT):
    """Mel spectrogram (features/mel.py): n_mels triangular Slaney-normalised filters on |STFT|^2,
    power_to_db(ref=max) -> /80 + 1."""
    _linear_is_power = True

    def __init__(self, sample_rate=16000, hop_length=512, decibels=True, n_mels=229, n_fft=2048, win_length=None,
                 center=True, htk=False, librosa_version='0.10', device='cuda:0'):
        super().__init__(sample_rate=sample_rate, hop_length=hop_length, decibels=decibels, win_length=win_length,
                         center=center, n_fft=n_fft, librosa_version=librosa_version, device=device)
        self.n_mels = n_mels
        self.htk = htk

    def _plan_args(self):
        return (int(self.sample_rate), int(self.n_fft), int(self.hop_length), int(self.win_length), int(self.n_mels),
                int(bool(self.htk)), int(bool(self.center)), self._pad_mode())

    def process_audio(self, audio):
        return self._process_host(audio, self.n_mels)

    def to_decibels(self, feats):
        """power_to_db(ref=np.max) on a host array (features/mel.py:78-96)."""
        S = np.asarray(feats)
        ref = S.max() if S.size else 1.0
        db = 10.0 * np.log10(np.maximum(1e-10, S)) - 10.0 * np.log10(np.maximum(1e-10, ref))
        return np.maximum(db, db.max() - 80.0) if S.size else db

    def get_feature_size(self):
        return self.n_mels

    def filterbank(self):
        """Dense (n_mels, n_fft//2+1) float32 copy of the plan's filterbank (for inspection/tests)."""
        out = np.zeros((self.n_mels, self.n_fft // 2 + 1), dtype=np.float32)
        _lib.check(_lib.lib().amtx_spec_filterbank(self._get_plan(), _lib.ptr(out)), 'amtx_spec_filterbank')
        return out


# ======================================================================================================
# CQT / VQT / HCQT / HVQT  (amt_tools/features/vqt.py, cqt.py, hvqt.py, hcqt.py)
# ======================================================================================================
C1_HZ = 32.70319566257483            # librosa.note_to_hz('C1')
_HANN_BANDWIDTH = 1.50018310546875   # librosa.filters.window_bandwidth('hann')


def _early_downsample_count(nyquist, filter_cutoff, hop_length, n_octaves):
    """librosa.core.constantq.__early_downsample_count, which the reference imports (features/vqt.py:7,95)."""
    c1 = max(0, int(np.ceil(np.log2(nyquist / filter_cutoff)) - 1) - 1)
    twos, h = 0, int(hop_length)
    while h > 0 and h % 2 == 0:
        twos, h = twos + 1, h // 2
    return min(c1, max(0, twos - n_octaves + 1))


class _CqtPlanOwner(object):
    """Lazily created amtx_cqt_plan shared by the VQT-family modules."""

    def _cqt_args(self):
        raise NotImplementedError

    def _device_index(self):
        return _index_of(self.device)

    def change_device(self, device):
        self.device = device
        for m in getattr(self, 'modules', []):
            m.device = device

    def _get_plan(self, device=None):
        """Per-device plan cache, see _SpecPlanOwner._get_plan."""
        import torch
        index = self._device_index() if device is None else _index_of(device)
        plans = self.__dict__.setdefault('_plans', {})
        plan = plans.get(index)
        if plan is None:
            if not torch.cuda.is_available():
                raise _lib.AmtxError('no GPU visible: the CQT front-end has no CPU fallback')
            fmin, harmonics, truncate = self._cqt_args()
            arr = (C.c_double * len(harmonics))(*[float(h) for h in harmonics])
            handle = C.c_void_p()
            with torch.cuda.device(index):
                _lib.check(_lib.lib().amtx_cqt_plan_create(C.byref(handle), int(self.sample_rate), int(self.hop_length), float(fmin),
                                                           int(self.n_bins), int(self.bins_per_octave), float(self.gamma), arr,
                                                           len(harmonics), int(truncate),
                                                           int(str(self.librosa_version).startswith('0.9'))), 'amtx_cqt_plan_create')
            plan = plans[index] = handle
        return plan

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_plans', None)
        state.pop('_workspace', None)
        return state

    def __del__(self):
        for plan in self.__dict__.get('_plans', {}).values():
            try:
                _lib.lib().amtx_cqt_plan_destroy(plan)
            except Exception:
                pass

    def process_batch(self, audio, model_layout=False):
        """(B, N) float32 CUDA tensor -> (B, C, F, T) float32 features on the device."""
        import torch
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
        audio = audio.contiguous()
        B, N = audio.shape
        L = _lib.lib()
        plan = self._get_plan(audio.device)
        T = _lib.check(L.amtx_cqt_num_frames(plan, N), 'amtx_cqt_num_frames')
        H = L.amtx_cqt_num_harmonics(plan)
        need = L.amtx_cqt_workspace_bytes(plan, B, N)
        ws = self.__dict__.get('_workspace')
        if ws is None or ws.numel() < need or ws.device != audio.device:
            self.__dict__['_workspace'] = None
            ws = self.__dict__['_workspace'] = _lib.alloc_workspace(need, audio.device)
        out = torch.empty((B, H, self.n_bins, T), dtype=torch.float32, device=audio.device)
        with torch.cuda.device(audio.device):
            _lib.check(L.amtx_cqt_forward(plan, _lib.ptr(audio), N, audio.stride(0) if B > 1 else N, B, int(bool(self.decibels)), _lib.ptr(ws), ws.numel(),
                                          _lib.ptr(out), _lib.current_stream(audio.device)), 'amtx_cqt_forward')
        return out

    def process_batch16(self, audio, split=False):
        """(B, N) float32 CUDA tensor -> (B, T, F, 8) bfloat16: the features of process_batch rounded to bf16 and laid out the way the
        Onsets & Frames engine's first conv kernel stages them (amtx_cqt_forward16: a position's harmonics in one 16-byte slot, slots
        C .. 7 zero) -- what OnsetsFrames.run_on_batch hands its engine when the model takes it (amtx_of_forward_feats16).
        split=True: (2, B, T, F, 8) -- plane 0 that map, plane 1 = bf16(feature - plane 0): the two planes the x3 engine multiplies with
        (amtx_cqt_forward16_split)."""
        import torch
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
        audio = audio.contiguous()
        B, N = audio.shape
        L = _lib.lib()
        plan = self._get_plan(audio.device)
        T = _lib.check(L.amtx_cqt_num_frames(plan, N), 'amtx_cqt_num_frames')
        assert L.amtx_cqt_num_harmonics(plan) <= 8
        need = L.amtx_cqt_workspace_bytes(plan, B, N)
        ws = self.__dict__.get('_workspace')
        if ws is None or ws.numel() < need or ws.device != audio.device:
            self.__dict__['_workspace'] = None
            ws = self.__dict__['_workspace'] = _lib.alloc_workspace(need, audio.device)
        out = torch.empty(((2,) if split else ()) + (B, T, self.n_bins, 8), dtype=torch.bfloat16, device=audio.device)
        with torch.cuda.device(audio.device):
            if split:
                _lib.check(L.amtx_cqt_forward16_split(plan, _lib.ptr(audio), N, audio.stride(0) if B > 1 else N, B, int(bool(self.decibels)), _lib.ptr(ws),
                                                      ws.numel(), _lib.ptr(out), out[0].numel(), _lib.current_stream(audio.device)), 'amtx_cqt_forward16_split')
            else:
                _lib.check(L.amtx_cqt_forward16(plan, _lib.ptr(audio), N, audio.stride(0) if B > 1 else N, B, int(bool(self.decibels)), _lib.ptr(ws), ws.numel(),
                                                _lib.ptr(out), _lib.current_stream(audio.device)), 'amtx_cqt_forward16')
        return out

    def _process_host(self, audio):
        import torch
        dev = torch.device('cuda', self._device_index())
        x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(dev).unsqueeze(0)
        return self.process_batch(x)[0].cpu().numpy()

    def frontend(self):
        from .models import SpectralFrontend
        return SpectralFrontend(self)


class VQT(FeatureModule, _CqtPlanOwner):
    """Variable-Q transform (features/vqt.py)."""

    def __init__(self, sample_rate=22050, hop_length=512, decibels=True, fmin=None, n_bins=84, bins_per_octave=12, gamma=None,
                 librosa_version='0.10', device='cuda:0'):
        FeatureModule.__init__(self, sample_rate, hop_length, 1, decibels)
        self.fmin = C1_HZ if fmin is None else fmin
        self.n_bins = n_bins
        self.bins_per_octave = bins_per_octave
        self.window = 'hann'
        # the reference keeps alpha in the librosa-0.9 convention and derives the default gamma from it (vqt.py:51-58)
        self.alpha = 2.0 ** (1.0 / self.bins_per_octave) - 1
        self.gamma = 24.7 * self.alpha / 0.108 if gamma is None else gamma
        self.n_octs = int(np.ceil(float(self.n_bins) / self.bins_per_octave))
        self.librosa_version = str(librosa_version)
        self.device = device

    def _cqt_args(self):
        return self.fmin, [1.0], False

    def get_early_ds_count(self):
        fmax = self.fmin * 2.0 ** ((self.n_bins - 1) / self.bins_per_octave)
        cQ = 1.0 / (2.0 ** (1. / self.bins_per_octave) - 1)
        freq_cutoff = fmax * (1 + 0.5 * (_HANN_BANDWIDTH / cQ)) + 0.5 * self.gamma
        return _early_downsample_count(self.sample_rate / 2.0, freq_cutoff, self.hop_length, self.n_octs)

    def get_expected_frames(self, audio):
        early = self.get_early_ds_count()
        k = np.arange(early, early + self.n_octs)
        sig_lens = np.ceil(len(audio) / (2 ** k))
        hop_lens = self.hop_length // (2 ** k)
        return int(min(sig_lens // hop_lens + 1))

    def get_sample_range(self, num_frames):
        factor = 2 ** self.get_early_ds_count()
        max_samples = ((num_frames * self.hop_length // factor) - 1) * factor
        min_samples = max(1, max_samples - self.hop_length + 1)
        return np.arange(min_samples, max_samples + 1)

    def process_audio(self, audio):
        return self._process_host(audio)

    def get_times(self, audio, at_start=False):
        times = super().get_times(audio)
        if at_start:
            # intended behaviour of vqt.py:217-225 (the reference floor-divides the tuple wavelet_lengths returns)
            alpha = self.alpha
            longest = (1.0 / alpha) * self.sample_rate / (self.fmin + self.gamma / alpha)
            times -= ((longest // 2) / self.sample_rate)
        return times

    def get_feature_size(self):
        return self.n_bins


class CQT(VQT):
    """Constant-Q transform = VQT with gamma = 0 (features/cqt.py)."""

    def __init__(self, sample_rate=22050, hop_length=512, decibels=True, fmin=None, n_bins=84, bins_per_octave=12,
                 librosa_version='0.10', device='cuda:0'):
        super().__init__(sample_rate, hop_length, decibels, fmin, n_bins, bins_per_octave, gamma=0,
                         librosa_version=librosa_version, device=device)


class HVQT(FeatureModule, _CqtPlanOwner):
    """Harmonic VQT (features/hvqt.py): one VQT per harmonic h * fmin, each dB-normalised on its own, truncated to the
    smallest expected frame count, stacked on the channel axis.  All harmonics run in one device pass: harmonics that
    are octaves apart share pyramid levels (and GEMM launches)."""

    def __init__(self, sample_rate=22050, hop_length=512, decibels=True, fmin=None, harmonics=None, n_bins=84, bins_per_octave=12,
                 gamma=None, librosa_version='0.10', device='cuda:0'):
        self.fmin = C1_HZ if fmin is None else fmin
        if harmonics is None:
            harmonics = [0.5, 1, 2, 3, 4, 5]
        harmonics.sort()
        self.harmonics = harmonics
        FeatureModule.__init__(self, sample_rate, hop_length, len(self.harmonics), decibels)
        self.modules = [VQT(sample_rate=sample_rate, hop_length=hop_length, decibels=decibels, fmin=h * self.fmin, n_bins=n_bins,
                            bins_per_octave=bins_per_octave, gamma=gamma, librosa_version=librosa_version, device=device)
                        for h in self.harmonics]
        self.n_bins = n_bins
        self.bins_per_octave = bins_per_octave
        self.gamma = self.modules[0].gamma
        self.librosa_version = str(librosa_version)
        self.device = device

    def _cqt_args(self):
        return self.fmin, self.harmonics, True

    def get_expected_frames(self, audio):
        return min(module.get_expected_frames(audio) for module in self.modules)

    def get_sample_range(self, num_frames):
        return self.modules[-1].get_sample_range(num_frames)

    def process_audio(self, audio):
        return self._process_host(audio)

    def get_times(self, audio, at_start=False):
        return self.modules[0].get_times(audio, at_start)[:self.get_expected_frames(audio)]

    def get_feature_size(self):
        return self.modules[0].get_feature_size()


class HCQT(HVQT):
    """Harmonic CQT (features/hcqt.py)."""

    def __init__(self, sample_rate=22050, hop_length=512, decibels=True, fmin=None, harmonics=None, n_bins=84, bins_per_octave=12,
                 librosa_version='0.10', device='cuda:0'):
        super().__init__(sample_rate, hop_length, decibels, fmin, harmonics, n_bins, bins_per_octave, gamma=0,
                         librosa_version=librosa_version, device=device)
