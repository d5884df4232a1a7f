"""
The reference's on-disk artefacts next to the hot path, so that a run that uses this package's
FeatureModules can share feature caches and labels with an amt-tools installation:

* feature cache files: `np.savez_compressed` with the keys `fs`, `hop_length`, `features`
  (amt_tools/datasets/common.py:236-270, tools/utils.py:3468-3502);
* ground-truth rasterisation: notes -> multi-pitch / onset / offset maps
  (amt_tools/tools/utils.py:620-700 filter_notes, :1665-1737 notes_to_multi_pitch, :2329-2378 notes_to_onsets,
  :2508-2552 notes_to_offsets).
"""
import os
import warnings

import numpy as np

from . import tools
from .transcribe import estimate_hop_length

__all__ = ['save_dict_npz', 'load_dict_npz', 'save_features', 'load_features', 'cached_process_audio', 'filter_notes',
           'notes_to_multi_pitch', 'notes_to_onsets', 'notes_to_offsets']


def save_dict_npz(path, d):
    np.savez_compressed(path, **d)


def load_dict_npz(path):
    return dict(np.load(path, allow_pickle=True))


def save_features(path, feats, fs, hop_length):
    """Write one track's features the way TranscriptionDataset.calculate_features does (datasets/common.py:259-266)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    save_dict_npz(path, {tools.KEY_FS: fs, tools.KEY_HOP: hop_length, tools.KEY_FEATS: feats})


def load_features(path):
    """(features, fs, hop_length) from a cache file (datasets/common.py:243-251); `features` may be None-valued when the
    feature module defers extraction to the model (a 0-d object array holding None)."""
    d = load_dict_npz(path)
    feats = d[tools.KEY_FEATS]
    feats = feats.item() if feats.size == 1 else feats
    return feats, d[tools.KEY_FS].item(), d[tools.KEY_HOP].item()


def cached_process_audio(data_proc, audio, path, sample_rate=None, hop_length=None, save=True):
    """Features of `audio` from the cache at `path` when it exists, else computed with `data_proc` (and cached when `save`):
    the get-or-compute step of TranscriptionDataset.calculate_features (datasets/common.py:236-276), including its warning
    when the cached hyper-parameters disagree with the expected ones."""
    if save and os.path.exists(path):
        feats, fs, hop = load_features(path)
    else:
        feats = data_proc.process_audio(audio)
        fs, hop = data_proc.get_sample_rate(), data_proc.get_hop_length()
        if save:
            save_features(path, feats, fs, hop)
    if (sample_rate is not None and sample_rate != fs) or (hop_length is not None and hop_length != hop):
        warnings.warn('Loaded features\' sampling rate or hop length differs from expected.', category=RuntimeWarning)
    return feats


def filter_notes(pitches, intervals, profile=None, min_time=-np.inf, max_time=np.inf, suppress_warnings=True):
    pitches_r = np.round(pitches)
    if profile is not None:
        in_pitch = np.logical_and(pitches_r >= profile.low, pitches_r <= profile.high)
        if np.sum(np.logical_not(in_pitch)) and not suppress_warnings:
            warnings.warn('Ignoring notes with nominal pitch exceeding supported boundaries.', category=RuntimeWarning)
    on_ok = intervals[:, 0] <= max_time
    off_ok = intervals[:, 1] >= min_time
    if not suppress_warnings:
        if np.sum(np.logical_not(on_ok)):
            warnings.warn('Ignoring notes with onsets occurring after specified time maximum.', category=RuntimeWarning)
        if np.sum(np.logical_not(off_ok)):
            warnings.warn('Ignoring notes with offsets occurring before specified time minimum.', category=RuntimeWarning)
    valid = np.logical_and(on_ok, off_ok)
    if profile is not None:
        valid = np.logical_and(valid, in_pitch)
    return pitches[valid], intervals[valid]


def notes_to_multi_pitch(pitches, intervals, times, profile, include_offsets=True):
    """(N,) MIDI pitches + (N,2) onset/offset seconds -> (F,T) float64 activation map on the frame grid `times`."""
    num_frames = len(times)
    multi_pitch = np.zeros((profile.get_range_len(), num_frames))
    _times = np.append(times, times[-1] + estimate_hop_length(times))
    pitches, intervals = filter_notes(pitches, intervals, profile, min_time=np.min(_times), max_time=np.max(_times))
    num_notes = len(pitches)
    pitches = np.round(pitches - profile.low).astype('int64')
    times_broadcast = np.concatenate([[_times]] * max(1, num_notes), axis=0)
    onsets = np.argmin(times_broadcast <= intervals[..., :1], axis=1) - 1
    offsets = np.argmin(times_broadcast <= intervals[..., 1:], axis=1) - 1
    onsets[onsets == -1], offsets[offsets == -1] = 0, num_frames - 1
    for i in range(num_notes):
        multi_pitch[pitches[i], onsets[i]: offsets[i] + int(include_offsets)] = 1
    return multi_pitch


def notes_to_onsets(pitches, intervals, times, profile, ambiguity=None):
    onset_times = np.copy(intervals[..., :1])
    offset_times = np.copy(intervals[..., 1:])
    if ambiguity is not None:
        offset_times = onset_times + np.minimum(offset_times - onset_times, ambiguity)
    else:
        offset_times = np.copy(onset_times)
    return notes_to_multi_pitch(pitches, np.concatenate((onset_times, offset_times), axis=-1), times, profile)


def notes_to_offsets(pitches, intervals, times, profile, ambiguity=None):
    offset_times = np.copy(intervals[..., 1:])
    onset_times = np.copy(offset_times)
    if ambiguity is not None:
        offset_times += ambiguity
    return notes_to_multi_pitch(pitches, np.concatenate((onset_times, offset_times), axis=-1), times, profile)
