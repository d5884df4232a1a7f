"""
ORACLE (test infrastructure only) -- NumPy restatement of the note decoding that
amt-tools' transcribe.py applies to the hot path's binary output.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (amt_tools_amd.*) never does.

What it restates (citations relative to /root/reference):
* amt_tools/tools/utils.py:369-471    multi_pitch_to_notes   (event walk; small pure-Python loop)
* amt_tools/tools/utils.py:2381-2412  multi_pitch_to_onsets
* amt_tools/tools/utils.py:3197-3229  estimate_hop_length
* amt_tools/tools/utils.py:2685-2746  sort_batched_notes / sort_notes (np.argsort, default kind)
* amt_tools/tools/utils.py:135-165    notes_to_batched_notes
* amt_tools/transcribe.py:420-481,722-763  (Stacked)NoteTranscriber.estimate with
  inhibition_window=None, minimum_duration=None (the paper scripts' settings)

Pinned bit-exactly against golden triples produced by the reference's own
NoteTranscriber (tools/gen_golden.py -> tests/golden/notes_*.npz).
"""

import numpy as np


def multi_pitch_to_onsets(multi_pitch):
    first_frame = multi_pitch[..., :1]
    adjacent_diff = multi_pitch[..., 1:] - multi_pitch[..., :-1]
    onsets = np.concatenate([first_frame, adjacent_diff], axis=-1)
    onsets[onsets <= 0] = 0
    return onsets


def estimate_hop_length(times):
    if not len(times):
        raise ValueError('Cannot estimate hop length from an empty time array.')
    times = np.sort(times)
    non_gaps = np.append([False], np.isclose(np.diff(times, n=2), 0))
    if not np.sum(non_gaps):
        raise ValueError('Time observations are too irregular.')
    return np.median(np.diff(times)[non_gaps])


def _sort_by_onset(pitches, intervals):
    batched = np.empty([0, 3])
    if len(pitches) > 0:
        batched = np.concatenate((intervals, np.expand_dims(pitches, axis=-1)), axis=-1)
    batched = batched[np.argsort(batched[..., 0])]
    return batched[..., 2], batched[:, :2]


def multi_pitch_to_notes(multi_pitch, times, low=21, onsets=None):
    """tools/utils.py:369-471 restated; `low` = profile.low (21 for the PianoProfile)."""
    if onsets is None:
        onsets = multi_pitch_to_onsets(multi_pitch)
    multi_pitch = np.logical_or(onsets, multi_pitch).astype('float32')
    onsets = multi_pitch_to_onsets(onsets)
    num_frames = multi_pitch.shape[-1]
    times = np.append(times, times[-1] + estimate_hop_length(times))
    pitches, intervals = list(), list()
    pitch_idcs, frame_idcs = onsets.nonzero()
    for pitch, frame in zip(pitch_idcs, frame_idcs):
        onset, offset = frame, frame + 1
        while offset != num_frames and multi_pitch[pitch, offset] and not onsets[pitch, offset]:
            offset += 1
        pitches.append(pitch + low)
        intervals.append([times[onset], times[offset]])
    pitches, intervals = np.array(pitches), np.array(intervals)
    return _sort_by_onset(pitches, intervals)


def note_transcriber(multi_pitch, onsets, times, low=21):
    """NoteTranscriber.estimate (transcribe.py:722-763) for one track: (88,T) x2 + times -> (K,3)
    rows [onset_s, offset_s, midi_pitch].  The reference argsorts by onset three times (inside
    multi_pitch_to_notes utils.py:469, in notes_to_stacked_notes utils.py:745 and in
    stacked_notes_to_notes utils.py:531); np.argsort's default kind is not stable, so all three passes
    are reproduced to keep the reference's row order among equal onsets."""
    pitches, intervals = multi_pitch_to_notes(np.array(multi_pitch, copy=True), times, low,
                                              None if onsets is None else np.array(onsets, copy=True))
    pitches, intervals = _sort_by_onset(pitches, intervals)
    pitches, intervals = _sort_by_onset(pitches, intervals)
    batched = np.empty([0, 3])
    if len(pitches) > 0:
        batched = np.concatenate((intervals, np.expand_dims(pitches, axis=-1)), axis=-1)
    return batched
