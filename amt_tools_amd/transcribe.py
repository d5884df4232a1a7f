"""
Note / pitch-list estimation for the hot path's output, mirroring the estimators of amt_tools/transcribe.py
that the paper scripts attach to Onsets & Frames (examples/papers/of_1.py:150-155):

* `NoteTranscriber.estimate(raw_output)`  -- amt_tools/transcribe.py:717-763 (+ StackedNoteTranscriber :420-481,
  tools.multi_pitch_to_notes tools/utils.py:369-471): binary (88,T) maps + times -> (K,3) rows
  [onset_s, offset_s, midi_pitch].  Bit-exact with the reference, including the row order among equal onsets
  (the reference argsorts by onset three times with NumPy's default, unstable sort; the same three passes run here).
  The per-event Python `while` walk is replaced by a vectorised next-stop scan (host arrays) or by the HIP kernel
  behind `amtx_notes_decode` (device tensors, whole batches: `decode_notes_batch`).
* `PitchListWrapper.estimate` -- amt_tools/transcribe.py:1038-1071, tools.multi_pitch_to_pitch_list utils.py:1023-1062.

The reference's own estimators keep working on this package's model output unchanged; these classes exist so a
batched offline driver (BASELINE config 5) does not spend its time in per-note Python loops.
"""

import numpy as np

from . import _lib, tools

__all__ = ['NoteTranscriber', 'PitchListWrapper', 'multi_pitch_to_notes', 'decode_notes_batch', 'estimate_hop_length', 'inhibit_activations']


def estimate_hop_length(times):
    """Spacing of a (possibly gappy) frame grid: the median of those steps that equal their predecessor, i.e. steps inside a
    regular run (behaviour contract: amt_tools/tools/utils.py:3197-3229, including the 1e-8 absolute tolerance of np.isclose
    against zero and ValueError for grids that are empty or have no two equal consecutive steps)."""
    grid = np.sort(np.asarray(times).ravel())            # dtype kept: run_offline hands float32 grids over (inference.py:36)
    if grid.size == 0:
        raise ValueError('hop length of an empty time grid is undefined')
    steps = grid[1:] - grid[:-1]
    change = steps[1:] - steps[:-1]                      # == np.diff(grid, n=2)
    regular = steps[1:][np.abs(change) <= 1e-8]          # a step counts when it repeats the step before it
    if regular.size == 0:
        raise ValueError('time grid has no regularly spaced stretch to take a hop length from')
    return np.median(regular)


def _impulses(x):
    """Positive first difference along time, first frame counts (tools/utils.py:2381-2412), as booleans."""
    x = np.asarray(x)
    first = x[..., :1] > 0
    return np.concatenate([first, (x[..., 1:] - x[..., :-1]) > 0], axis=-1)


def _sort_by_onset(batched):
    return batched[np.argsort(batched[..., 0])]


def _extend_times(times):
    """The frame grid plus one more frame (tools/utils.py:441-442): offsets may point one past the last frame."""
    return np.append(times, times[-1] + estimate_hop_length(times))


def _events_to_notes(pitch_idcs, on_frames, off_frames, times, low, times_ext=None, min_duration=None):
    """(key, onset frame, offset frame) events in np.nonzero order -> (K,3) rows ordered like the reference.
    `times_ext` = _extend_times(times) when the caller already has it (one grid shared by a whole batch).
    `min_duration`: the reference's duration filter (transcribe.py:36-80), which sits between its first and second sort:
    notes shorter than the threshold go (threshold 0: zero-length notes go)."""
    if times_ext is None:
        times_ext = _extend_times(times)
    if len(pitch_idcs) == 0:
        return np.empty([0, 3])
    # sort_notes in multi_pitch_to_notes (utils.py:469), notes_to_stacked_notes (:745), stacked_notes_to_notes (:531): three
    # successive argsorts by onset with NumPy's default (unstable) sort.  The same three argsorts run here on the 1-D onset
    # column only and their permutations are composed; the (K,3) rows are gathered once (same result as re-indexing the whole
    # array three times, 40 % less host time per clip -- the host assembly bounds the batched transcription driver).
    # float64 keys whatever the grid's dtype: the reference's sort runs on the (K, 3) array np.concatenate built from the float32 / float64
    # intervals and the int64 pitches, i.e. on float64 (utils.py:135-164) -- and NumPy's unstable sort orders ties differently per key width
    onset_t = times_ext[on_frames].astype(np.float64)
    perm = np.argsort(onset_t)
    if min_duration is not None:
        dur = times_ext[np.asarray(off_frames)[perm]] - onset_t[perm]
        perm = perm[dur >= min_duration] if min_duration else perm[dur > min_duration]
        if len(perm) == 0:
            return np.empty([0, 3])
    for _ in range(2):
        perm = perm[np.argsort(onset_t[perm])]
    batched = np.empty((len(perm), 3))
    batched[:, 0] = onset_t[perm]
    batched[:, 1] = times_ext[np.asarray(off_frames)[perm]]
    batched[:, 2] = np.asarray(pitch_idcs)[perm] + low
    return batched


def inhibit_activations(activations, times, window_length):
    """Keep an activation only if no KEPT activation of the same row started less than `window_length` seconds before it
    (behaviour contract: amt_tools/tools/utils.py:2987-3038, which rescans the whole map once per kept activation).  Per row the
    non-zero frames are walked once: keep the first, jump with one binary search to the first frame at or past its time +
    window, keep the next non-zero from there on.  Returns a new {0,1} map of the input's dtype."""
    activations = np.asarray(activations)
    times = np.asarray(times)
    out = np.zeros_like(activations)
    rows, frames = np.nonzero(activations)
    bounds = np.searchsorted(rows, np.arange(activations.shape[0] + 1))
    for r in range(activations.shape[0]):
        fr = frames[bounds[r]:bounds[r + 1]]
        i = 0
        while i < len(fr):
            out[r, fr[i]] = 1
            release = np.searchsorted(times, times[fr[i]] + window_length, side='left')      # first frame outside the window
            i = np.searchsorted(fr, max(release, fr[i] + 1), side='left')
    return out


def multi_pitch_to_notes(multi_pitch, times, low=tools.DEFAULT_PIANO_LOWEST_PITCH, onsets=None, min_duration=None):
    """Vectorised tools.multi_pitch_to_notes for host arrays: returns (K,3) batched notes."""
    multi_pitch = np.asarray(multi_pitch)
    if onsets is None:
        imp = _impulses(multi_pitch)
        active = multi_pitch != 0
    else:
        onsets = np.asarray(onsets)
        imp = _impulses(onsets)
        active = np.logical_or(onsets, multi_pitch)
    T = multi_pitch.shape[-1]
    stop = np.logical_or(~active, imp)
    idx = np.where(stop, np.arange(T), T)
    nearest = np.minimum.accumulate(idx[..., ::-1], axis=-1)[..., ::-1]          # nearest stop at >= t
    after = np.concatenate([nearest[..., 1:], np.full(nearest.shape[:-1] + (1,), T)], axis=-1)   # strictly after t
    pitch_idcs, frame_idcs = imp.nonzero()
    return _events_to_notes(pitch_idcs, frame_idcs, after[pitch_idcs, frame_idcs], np.asarray(times), low, min_duration=min_duration)


from ._order_pool import order_batch, reference_order as _reference_order     # noqa: E402  (the reference's three unstable argsorts)


_GRIDS = {}        # (device, shape, bytes) -> the extended time grid on the device
_D2H_STREAMS = {}  # device -> the stream note rows are copied to the host on


def _grid_on_device(ext, dev):
    """The extended time grid as a device tensor, uploaded once per distinct grid.  The upload is a copy from pageable memory on the
    current stream: it returns only when every kernel enqueued before it has finished, so one upload per batch made the batched driver wait
    for its own model forward before it could go on to the previous batch's host work (4 x 8 ms per 2048 clips; config 5's host-to-host
    rate went from 14 to 25 M frames/s with the grid cached)."""
    import hashlib
    import torch
    # keyed on a digest, not on the bytes themselves (a (B, T) grid is megabytes per batch)
    key = (str(dev), ext.shape, ext.dtype.str, hashlib.blake2b(np.ascontiguousarray(ext).view(np.uint8).reshape(-1), digest_size=16).digest())
    t = _GRIDS.get(key)
    if t is None:
        if len(_GRIDS) >= 16:
            _GRIDS.pop(next(iter(_GRIDS)))         # the oldest ONE; its memory is only reused behind the streams recorded below
        t = _GRIDS[key] = torch.from_numpy(ext).to(dev)
    # the decoder's kernels read the grid on the caller's current stream, which need not be the stream it was allocated on: tell the
    # caching allocator, so that an evicted grid is not handed out again while those kernels are still queued
    t.record_stream(torch.cuda.current_stream(t.device))
    return t


class _PendingNotes(object):
    """Device half of the decoder already enqueued (amtx_notes_decode + amtx_notes_rows: one dense (E,3) float64 array of note rows in
    np.nonzero order per clip, its onset column and the per-clip offsets); `result()` copies them to the host and applies the
    reference's row order clip by clip.  Lets a caller enqueue the next batch's kernels before paying for this batch's host work."""

    def __init__(self, rows, onset_col, offsets, B, retry):
        self._rows, self._onset, self._offsets, self._B, self._retry = rows, onset_col, offsets, B, retry
        import torch
        self._done = torch.cuda.Event()
        self._done.record(torch.cuda.current_stream(rows.device))

    def result(self):
        import torch
        # The copies to the host run on their own stream behind this batch's event: on the caller's stream they would queue behind
        # whatever the caller has enqueued since (the batched driver: the NEXT batch's whole forward pass)
        dev = self._rows.device
        side = _D2H_STREAMS.get(str(dev))
        if side is None:
            side = _D2H_STREAMS[str(dev)] = torch.cuda.Stream(dev)
        side.wait_event(self._done)
        with torch.cuda.stream(side):
            off = self._offsets.cpu().numpy()
            total = int(off[-1])
            if total > self._rows.shape[0]:             # more notes than the first buffer held: once more with the exact size
                again = self._retry(total)              # enqueued on this side stream, behind the decoder's event
                self._rows, self._onset, off = again._rows, again._onset, again._offsets.cpu().numpy()
            rows = self._rows[:total].cpu().numpy()
            onset = self._onset[:total].cpu().numpy()
        # the reference's row order per clip: NumPy's own argsort, three times (amt_tools_amd/_order_pool.py: a few helper processes for
        # whole batches, in-process for small ones)
        return order_batch(rows, onset, off, self._B)


def decode_notes_batch_async(onsets, multi_pitch, times, low=tools.DEFAULT_PIANO_LOWEST_PITCH, rows_capacity=None):
    """Enqueue the device decoder and return a handle; `handle.result()` -> list of B (K,3) float64 arrays.  (B,88,T) fp32 CUDA tensors
    (onsets may be None); `times` is one (T,) grid or a (B,T) array.  Two C-ABI calls: amtx_notes_decode (event walk per key) and
    amtx_notes_rows (compaction into the reference's batched-notes rows, frames -> seconds through the extended grid)."""
    import torch
    assert multi_pitch.is_cuda and multi_pitch.dim() == 3
    B, K, T = multi_pitch.shape
    dev = multi_pitch.device
    multi_pitch = multi_pitch.contiguous().float()
    if onsets is not None:
        onsets = onsets.contiguous().float()
    times = np.asarray(times)
    if times.ndim == 1:
        ext = _extend_times(times).astype(np.float64)                  # float32 grids (run_offline) convert exactly, as in the reference's float64 rows
        stride = 0
    else:
        ext = np.stack([_extend_times(t) for t in times]).astype(np.float64)
        stride = ext.shape[1]
    assert ext.shape[-1] == T + 1
    ext_d = _grid_on_device(np.ascontiguousarray(ext), dev)
    cap = T // 2 + 2
    pairs = torch.empty((B * K, cap, 2), dtype=torch.int32, device=dev)
    counts = torch.empty((B * K,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    with torch.cuda.device(dev):
        _lib.check(L.amtx_notes_decode(_lib.ptr(onsets), _lib.ptr(multi_pitch), B, K, T, cap, _lib.ptr(pairs), _lib.ptr(counts),
                                       _lib.current_stream(dev)), 'amtx_notes_decode')

    def rows_pass(capacity):
        rows = torch.empty((capacity, 3), dtype=torch.float64, device=dev)
        onset_col = torch.empty((capacity,), dtype=torch.float64, device=dev)
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.amtx_notes_rows(_lib.ptr(pairs), _lib.ptr(counts), B, K, cap, _lib.ptr(ext_d), stride, int(low), _lib.ptr(rows),
                                         _lib.ptr(onset_col), capacity, _lib.ptr(offsets), _lib.current_stream(dev)), 'amtx_notes_rows')
        return _PendingNotes(rows, onset_col, offsets, B, rows_pass)

    # 1024 notes per clip on average covers any realistic transcription (the synthetic bench clips decode to ~400); a denser batch is
    # caught in result() through the total the device reports and decoded again into a buffer of the exact size
    return rows_pass(int(rows_capacity) if rows_capacity else max(B * 1024, 1 << 14))


def decode_notes_batch(onsets, multi_pitch, times, low=tools.DEFAULT_PIANO_LOWEST_PITCH):
    """Device path: (B,88,T) fp32 CUDA tensors (onsets may be None) -> list of B (K,3) float64 arrays.
    `times` is one (T,) grid shared by the batch or a (B,T) array."""
    return decode_notes_batch_async(onsets, multi_pitch, times, low).result()


class NoteTranscriber(object):
    """amt_tools.transcribe.NoteTranscriber (transcribe.py:717-785 over StackedNoteTranscriber :373-481), all constructor options:
    `inhibition_window` (seconds after a kept onset in which the same pitch cannot start again) acts -- exactly as in the
    reference, transcribe.py:463-468 -- only when the model supplies no onset map (the onsets are then the inhibited positive
    first difference of the multi-pitch map); `minimum_duration` drops shorter notes (0: zero-length notes)."""

    def __init__(self, profile, inhibition_window=None, minimum_duration=None, multi_pitch_key=None, onsets_key=None,
                 offsets_key=None, estimates_key=None, save_dir=None):
        self.inhibition_window = inhibition_window
        self.minimum_duration = minimum_duration
        self.profile = profile
        self.multi_pitch_key = tools.KEY_MULTIPITCH if multi_pitch_key is None else multi_pitch_key
        self.onsets_key = tools.KEY_ONSETS if onsets_key is None else onsets_key
        self.estimates_key = tools.KEY_NOTES if estimates_key is None else estimates_key
        self.save_dir = save_dir

    @staticmethod
    def get_default_key():
        return tools.KEY_NOTES

    def get_key(self):
        return self.estimates_key

    def estimate(self, raw_output):
        multi_pitch = tools.tensor_to_array(tools.unpack_dict(raw_output, self.multi_pitch_key))
        onsets = tools.tensor_to_array(tools.unpack_dict(raw_output, self.onsets_key))
        times = tools.tensor_to_array(tools.unpack_dict(raw_output, tools.KEY_TIMES))
        if self.inhibition_window is not None and onsets is None:
            derived = _impulses(multi_pitch).astype(np.asarray(multi_pitch).dtype)
            onsets = inhibit_activations(derived, times, self.inhibition_window)
        return multi_pitch_to_notes(multi_pitch, times, self.profile.low, onsets, self.minimum_duration)

    def process_track(self, raw_output, track=None):
        return {self.get_key(): self.estimate(raw_output)}


class PitchListWrapper(object):
    def __init__(self, profile, multi_pitch_key=None, estimates_key=None, save_dir=None):
        self.profile = profile
        self.multi_pitch_key = tools.KEY_MULTIPITCH if multi_pitch_key is None else multi_pitch_key
        self.estimates_key = tools.KEY_PITCHLIST if estimates_key is None else estimates_key

    def get_key(self):
        return self.estimates_key

    def estimate(self, raw_output):
        multi_pitch = tools.tensor_to_array(tools.unpack_dict(raw_output, self.multi_pitch_key))
        times = tools.tensor_to_array(tools.unpack_dict(raw_output, tools.KEY_TIMES))
        num_frames = multi_pitch.shape[-1]
        pitch_list = [np.empty(0)] * num_frames
        for i in np.where(np.sum(multi_pitch, axis=-2) > 0)[-1]:
            pitch_list[i] = (self.profile.low + np.where(multi_pitch[..., i])[-1]).astype('float')
        return times, pitch_list

    def process_track(self, raw_output, track=None):
        return {self.get_key(): self.estimate(raw_output)}
