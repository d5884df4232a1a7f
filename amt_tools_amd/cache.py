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
    """Drop notes that cannot be drawn: nominal (rounded) pitch outside the instrument's range, onset after `max_time`, or offset
    before `min_time`.  Returns (pitches, intervals) of the survivors, order kept.  Behaviour contract:
    amt_tools/tools/utils.py:620-700."""
    pitches, intervals = np.asarray(pitches), np.asarray(intervals)
    reasons = []
    if profile is not None:
        nominal = np.round(pitches)
        reasons.append(('pitch outside the supported range', (nominal < profile.low) | (nominal > profile.high)))
    reasons.append(('onset after the time maximum', intervals[:, 0] > max_time))
    reasons.append(('offset before the time minimum', intervals[:, 1] < min_time))
    drop = np.zeros(len(pitches), dtype=bool)
    for why, mask in reasons:
        if not suppress_warnings and mask.any():
            warnings.warn(f'Ignoring {int(mask.sum())} note(s): {why}.', category=RuntimeWarning)
        drop |= mask
    return pitches[~drop], intervals[~drop]


def _frame_of(grid_ext, t, overflow):
    """Index of the last grid point <= t on the extended (T+1)-point grid; times before the first point and times at or beyond
    the extension point map to `overflow` (the reference's `argmin(...) - 1 == -1` case covers both, utils.py:1716-1722)."""
    idx = np.searchsorted(grid_ext, t, side='right') - 1
    last = len(grid_ext) - 1
    return np.where((idx < 0) | (idx >= last), overflow, idx)


def notes_to_multi_pitch(pitches, intervals, times, profile, include_offsets=True):
    """(N,) MIDI pitches + (N,2) onset/offset seconds -> (keys, T) float64 activation map on the ascending frame grid `times`.
    Behaviour contract: amt_tools/tools/utils.py:1665-1737.  Every note becomes +1 / -1 in a per-key difference array at its
    first frame / one past its last frame; a running sum along time then marks the covered frames -- O(N + keys*T) with no
    per-note slice assignment and no (N, T) broadcast."""
    times = np.asarray(times)
    T = len(times)
    keys = profile.get_range_len()
    grid_ext = np.append(times, times[-1] + estimate_hop_length(times))
    pitches, intervals = filter_notes(pitches, intervals, profile, min_time=grid_ext.min(), max_time=grid_ext.max())
    if len(pitches) == 0:
        return np.zeros((keys, T))
    key = np.round(pitches - profile.low).astype(np.int64)
    first = _frame_of(grid_ext, intervals[:, 0], 0)
    stop = np.minimum(_frame_of(grid_ext, intervals[:, 1], T - 1) + int(include_offsets), T)      # exclusive
    drawn = stop > first
    delta = np.zeros((keys, T + 1), dtype=np.int64)
    np.add.at(delta, (key[drawn], first[drawn]), 1)
    np.add.at(delta, (key[drawn], stop[drawn]), -1)
    return (np.cumsum(delta[:, :T], axis=1) > 0).astype(np.float64)


def notes_to_onsets(pitches, intervals, times, profile, ambiguity=None):
    """Onset map: every note shrunk to its onset frame, or to min(duration, `ambiguity`) seconds from its onset
    (behaviour contract: amt_tools/tools/utils.py:2329-2378)."""
    start = np.asarray(intervals, dtype=np.float64)[:, 0]
    length = np.zeros_like(start) if ambiguity is None else np.minimum(np.asarray(intervals, dtype=np.float64)[:, 1] - start, ambiguity)
    return notes_to_multi_pitch(pitches, np.stack([start, start + length], axis=1), times, profile)


def notes_to_offsets(pitches, intervals, times, profile, ambiguity=None):
    """Offset map: every note shrunk to its offset frame, optionally extended by `ambiguity` seconds
    (behaviour contract: amt_tools/tools/utils.py:2508-2552)."""
    end = np.asarray(intervals, dtype=np.float64)[:, 1]
    tail = 0.0 if ambiguity is None else ambiguity
    return notes_to_multi_pitch(pitches, np.stack([end, end + tail], axis=1), times, profile)
