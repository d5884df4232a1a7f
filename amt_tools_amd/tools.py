"""
Host-side contract pieces of amt_tools.tools that the hot path touches: dictionary keys
(amt_tools/tools/constants.py:46-69,118-137), the PianoProfile (amt_tools/tools/instrument.py:13-100),
dict plumbing with the reference's semantics (amt_tools/tools/utils.py:3505-3745) but without its
repeated whole-batch deep copies, activation helpers and RMS normalisation.
"""

import numpy as np

try:
    import torch
except Exception:   # pragma: no cover
    torch = None

# ---- dictionary keys / constants (values are the contract, amt_tools/tools/constants.py) ----------
KEY_TRACK = 'track'
KEY_AUDIO = 'audio'
KEY_FS = 'fs'
KEY_HOP = 'hop_length'
KEY_FEATS = 'features'
KEY_MULTIPITCH = 'multi_pitch'
KEY_PITCHLIST = 'pitch_list'
KEY_ONSETS = 'onsets'
KEY_OFFSETS = 'offsets'
KEY_TIMES = 'times'
KEY_TABLATURE = 'tablature'
KEY_NOTES = 'notes'
KEY_OUTPUT = 'model_output'
KEY_LOSS = 'loss'
KEY_LOSS_TOTAL = 'loss_total'
KEY_LOSS_ONSETS = 'loss_onsets'
KEY_LOSS_OFFSETS = 'loss_offsets'
KEY_LOSS_PITCH = 'loss_pitch'
DEFAULT_PIANO_LOWEST_PITCH = 21
DEFAULT_PIANO_HIGHEST_PITCH = 108
DEFAULT_GUITAR_LABELS = ['E', 'A', 'D', 'G', 'B', 'e']
DEFAULT_GUITAR_TUNING = ['E2', 'A2', 'D3', 'G3', 'B3', 'E4']
DEFAULT_GUITAR_NUM_FRETS = 19
FLOAT32 = 'float32'
FLOAT64 = 'float64'
PYT_MODEL = 'model'
PYT_STATE = 'opt-state'
PYT_EXT = 'pt'


class InstrumentProfile(object):
    def __init__(self, low, high):
        self.low = low
        self.high = high

    def get_midi_range(self):
        return np.arange(self.low, self.high + 1)

    def get_range_len(self):
        return self.high - self.low + 1


class PianoProfile(InstrumentProfile):
    """88 keys, MIDI 21..108 (amt_tools/tools/instrument.py:65-100)."""

    def __init__(self, low=None, high=None):
        super().__init__(DEFAULT_PIANO_LOWEST_PITCH if low is None else low,
                         DEFAULT_PIANO_HIGHEST_PITCH if high is None else high)

    def get_num_dofs(self):
        return 1


def note_to_midi(note):
    """'E2' -> 40, 'A#3' / 'Bb3' -> 58 (scientific pitch notation, C4 = 60): what librosa.note_to_midi returns for the plain
    note names the instrument profiles use (amt_tools/tools/instrument.py:148-160)."""
    if isinstance(note, (list, tuple)):
        return [note_to_midi(n) for n in note]
    pc = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}[note[0].upper()]
    i = 1
    while i < len(note) and note[i] in '#b':
        pc += 1 if note[i] == '#' else -1
        i += 1
    octave = int(note[i:]) if i < len(note) else 0
    return 12 * (octave + 1) + pc


class TablatureProfile(InstrumentProfile):
    """Instruments with several degrees of freedom, e.g. strings (amt_tools/tools/instrument.py:103-260)."""

    def __init__(self, tuning, num_pitches):
        self.tuning = tuning
        self.num_pitches = num_pitches
        midi_tuning = self.get_midi_tuning()
        super().__init__(midi_tuning[0], midi_tuning[-1] - 1 + self.num_pitches)

    def get_num_dofs(self):
        return len(self.tuning)

    def get_midi_tuning(self):
        return note_to_midi(list(self.tuning))

    def get_dof_midi_range(self):
        tuning = self.get_midi_tuning()
        return np.array([np.arange(tuning[i], tuning[i] + self.num_pitches) for i in range(self.get_num_dofs())])

    def get_fret(self, midi_pitch, string):
        return midi_pitch - self.get_midi_tuning()[string]

    def get_pitch(self, string, fret):
        return self.get_midi_tuning()[string] + fret


class GuitarProfile(TablatureProfile):
    """Six strings in standard tuning, 19 frets by default (amt_tools/tools/instrument.py:263-305)."""

    def __init__(self, tuning=None, num_frets=None):
        if tuning is None:
            tuning = DEFAULT_GUITAR_TUNING
        if num_frets is None:
            num_frets = DEFAULT_GUITAR_NUM_FRETS
        super().__init__(tuning, num_frets + 1)

    def get_num_frets(self):
        return self.num_pitches - 1


# ---- dict plumbing ---------------------------------------------------------------------------------
def unpack_dict(data, key):
    """Entry for `key`, or None (amt_tools/tools/utils.py:3823-3853)."""
    return data[key] if (isinstance(data, dict) and key in data) else None


def query_dict(dictionary, key):
    return isinstance(dictionary, dict) and key in dictionary.keys()


def _map_dict(track, fn):
    """New dict (recursively), every leaf passed through fn; the caller's dict is never mutated."""
    out = dict()
    for k, v in track.items():
        out[k] = _map_dict(v, fn) if isinstance(v, dict) else fn(v)
    return out


def dict_to_device(track, device):
    return _map_dict(track, lambda v: v.to(device) if torch is not None and isinstance(v, torch.Tensor) else v)


def dict_to_dtype(track, dtype):
    return _map_dict(track, lambda v: v.astype(dtype) if isinstance(v, np.ndarray) else v)


def dict_to_tensor(track):
    return _map_dict(track, lambda v: torch.from_numpy(v) if isinstance(v, np.ndarray) else v)


def array_to_tensor(data, device=None):
    """amt_tools/tools/utils.py:3438-3465."""
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(data)
        if device is not None:
            data = data.to(device)
    return data


def framify_activations(activations, win_length, hop_length=1, pad=True):
    """Overlapping windows along the last axis: (..., T) -> (..., T', win_length), centre zero padding by win_length // 2 when
    `pad` (amt_tools/tools/utils.py:2922-2984; librosa.util.pad_center = symmetric zero padding, extra sample on the right)."""
    num_frames = activations.shape[-1]
    pad_length = win_length // 2
    num_frames_ = num_frames + 2 * pad_length if pad else max(win_length, num_frames)
    lpad = (num_frames_ - num_frames) // 2
    widths = [(0, 0)] * (activations.ndim - 1) + [(lpad, num_frames_ - num_frames - lpad)]
    activations = np.pad(activations, widths, mode='constant')
    num_hops = (num_frames_ - 2 * pad_length) // hop_length
    chunks = [np.expand_dims(activations[..., i: i + win_length], axis=-2) for i in np.arange(0, num_hops) * hop_length]
    return np.concatenate(chunks, axis=-2)


def tensor_to_array(data):
    if torch is not None and isinstance(data, torch.Tensor):
        data = data.cpu().detach().numpy()
    return data


def dict_to_array(track):
    return _map_dict(track, tensor_to_array)


def dict_unsqueeze(track, dim=0):
    def f(v):
        if torch is not None and isinstance(v, torch.Tensor):
            return v.unsqueeze(dim)
        if isinstance(v, np.ndarray):
            return np.expand_dims(v, axis=dim)
        return v
    return _map_dict(track, f)


def dict_squeeze(track, dim=None):
    def f(v):
        if (torch is not None and isinstance(v, torch.Tensor)) or isinstance(v, np.ndarray):
            if dim is None:
                return v.squeeze()
            if v.ndim > 0 and v.shape[dim] == 1:
                return v.squeeze(dim)
        return v
    out = dict()
    for k, v in track.items():
        # the reference recurses into nested dicts WITHOUT forwarding `dim` (utils.py:3681-3683)
        out[k] = dict_squeeze(v) if isinstance(v, dict) else f(v)
    return out


# ---- activation helpers ----------------------------------------------------------------------------
def threshold_activations(activations, threshold=0.5):
    """In place, like the reference (amt_tools/tools/utils.py:2896-2919): < thr -> 0, everything else -> 1."""
    activations[activations < threshold] = 0
    activations[activations != 0] = 1
    return activations


def multi_pitch_to_onsets(multi_pitch):
    """Positive first difference along time, first frame counts (amt_tools/tools/utils.py:2381-2412).
    Accepts ndarray or tensor; returns the same kind."""
    if torch is not None and isinstance(multi_pitch, torch.Tensor):
        onsets = torch.cat([multi_pitch[..., :1], multi_pitch[..., 1:] - multi_pitch[..., :-1]], dim=-1)
        return torch.clamp(onsets, min=0)
    onsets = np.concatenate([multi_pitch[..., :1], multi_pitch[..., 1:] - multi_pitch[..., :-1]], axis=-1)
    onsets[onsets <= 0] = 0
    return onsets


def multi_pitch_to_offsets(multi_pitch):
    """Where activity ceases; pitches active in the last frame count (amt_tools/tools/utils.py:2555-2589).
    Accepts ndarray or tensor; returns the same kind."""
    if torch is not None and isinstance(multi_pitch, torch.Tensor):
        offsets = torch.cat([multi_pitch[..., :-1] - multi_pitch[..., 1:], multi_pitch[..., -1:]], dim=-1)
        return torch.clamp(offsets, min=0)
    offsets = np.concatenate([multi_pitch[..., :-1] - multi_pitch[..., 1:], multi_pitch[..., -1:]], axis=-1)
    offsets[offsets <= 0] = 0
    return offsets


def rms_norm(audio):
    """Root-mean-square normalisation (amt_tools/tools/utils.py:2789-2814)."""
    rms = np.sqrt(np.mean(audio ** 2))
    if rms > 0:
        audio = audio / rms
    return audio


def rms_norm_batch(audio):
    """rms_norm for a (B, N) float32 CUDA tensor of clips, on the device (amtx_rms_norm)."""
    from . import _lib
    assert audio.is_cuda and audio.dim() == 2 and audio.dtype == torch.float32
    audio = audio.contiguous()
    B, N = audio.shape
    L = _lib.lib()
    ws = torch.empty(int(L.amtx_rms_norm_workspace_bytes(B, N)), dtype=torch.uint8, device=audio.device)
    out = torch.empty_like(audio)
    with torch.cuda.device(audio.device):
        # a one-clip batch may carry any stride in its size-1 dimension (numpy's x[None] has 0): the rows are N apart by definition then
        _lib.check(L.amtx_rms_norm(_lib.ptr(audio), N, audio.stride(0) if B > 1 else N, B, _lib.ptr(out), out.stride(0) if B > 1 else N, _lib.ptr(ws), ws.numel(),
                                   _lib.current_stream(audio.device)), 'amtx_rms_norm')
    return out


def seed_everything(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    if torch is not None:
        torch.manual_seed(seed)


def register_safe_globals():
    """amt_tools/train.py:106 resumes with `torch.load(model_path)` of a whole pickled module; torch >= 2.6 defaults to
    weights_only=True there, which rejects any class that is not allow-listed (the reference's own classes fail the same way).
    Call this once before `train(..., resume=True)`: it registers every class a pickled model of this package can contain
    (the package's modules, feature modules and profiles, and the stock torch.nn layers they are built from)."""
    from torch import nn
    from . import features, models
    classes = [getattr(models, n) for n in ('TranscriptionModel', 'OutputLayer', 'LogisticBank', 'SoftmaxGroups', 'AcousticModel',
                                            'LanguageModel', 'OnsetsFrames', 'OnsetsFrames2', 'TabCNN', 'SpectralFrontend')]
    classes += [getattr(features, n) for n in features.__all__]
    classes += [InstrumentProfile, PianoProfile, TablatureProfile, GuitarProfile]
    classes += [nn.Sequential, nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d, nn.Dropout, nn.Linear, nn.LSTM, nn.SyncBatchNorm]
    torch.serialization.add_safe_globals(classes)
    return classes
