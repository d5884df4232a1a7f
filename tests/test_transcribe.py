"""Note decoding of the product (vectorised host path and HIP kernel) bit-exact against the reference
NoteTranscriber's recorded output (tests/golden/notes_*.npz) and against the oracle on random maps."""
import numpy as np
import pytest

from conftest import load_golden
from amt_tools_amd import tools
from amt_tools_amd.transcribe import NoteTranscriber, PitchListWrapper, multi_pitch_to_notes
from oracle import notes_np

CASES = ['notes_dense.npz', 'notes_sparse.npz', 'notes_noonsets.npz', 'notes_empty.npz', 'notes_f32times.npz']


@pytest.mark.parametrize('name', CASES)
def test_host_decoder_bit_exact_with_reference(name):
    g = load_golden(name)
    raw = {tools.KEY_MULTIPITCH: g['multi_pitch'].copy(), tools.KEY_TIMES: g['times'].copy()}
    if int(g['with_onsets']):
        raw[tools.KEY_ONSETS] = g['onsets'].copy()
    notes = NoteTranscriber(tools.PianoProfile()).estimate(raw)
    assert notes.shape == g['notes'].shape and np.array_equal(notes, g['notes'])
    np.testing.assert_array_equal(raw[tools.KEY_MULTIPITCH], g['multi_pitch'])      # inputs not modified


@pytest.mark.parametrize('name', ['notes_inhibit.npz', 'notes_inhibit_onsets.npz', 'notes_mindur.npz', 'notes_mindur0.npz', 'notes_inhibit_mindur.npz'])
def test_note_transcriber_options_bit_exact_with_reference(name):
    """inhibition_window / minimum_duration (transcribe.py:373-481): goldens recorded from the reference's NoteTranscriber with the
    same options, including its quirk that the inhibition window is ignored when an onset map is supplied."""
    g = load_golden(name)
    iw, md = float(g['inhibition_window']), float(g['minimum_duration'])
    est = NoteTranscriber(tools.PianoProfile(), inhibition_window=None if iw < 0 else iw, minimum_duration=None if md < 0 else md)
    raw = {tools.KEY_MULTIPITCH: g['multi_pitch'].copy(), tools.KEY_TIMES: g['times'].copy()}
    if int(g['with_onsets']):
        raw[tools.KEY_ONSETS] = g['onsets'].copy()
    notes = est.estimate(raw)
    assert notes.shape == g['notes'].shape and np.array_equal(notes, g['notes'])
    np.testing.assert_array_equal(raw[tools.KEY_MULTIPITCH], g['multi_pitch'])


def test_inhibit_activations_is_a_greedy_per_row_scan():
    from amt_tools_amd.transcribe import inhibit_activations
    times = np.arange(10) * 0.1
    act = np.array([[1, 1, 0, 1, 0, 0, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 1]], dtype=np.float32)
    out = inhibit_activations(act, times, 0.25)          # window covers the kept frame and the next two
    np.testing.assert_array_equal(out, [[1, 0, 0, 1, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 1]])
    np.testing.assert_array_equal(inhibit_activations(act, times, 0.0), act)     # an empty window still keeps everything


def test_host_decoder_random_maps_vs_oracle():
    rng = np.random.default_rng(0)
    for T in (1, 2, 63, 64, 65, 200):
        mp = (rng.random((88, T)) < 0.3).astype(np.float32)
        on = (rng.random((88, T)) < 0.1).astype(np.float32)
        times = np.arange(T) * 512 / 22050.0 if T > 2 else np.arange(3)[:T] * 0.5
        if T < 3:
            continue                                             # the reference cannot estimate a hop from < 3 frames
        for o in (on, None):
            ref = notes_np.note_transcriber(mp, o, times)
            got = multi_pitch_to_notes(mp, times, 21, o)
            assert np.array_equal(got, ref)


def test_host_decoder_float32_grid_with_many_chords_orders_like_the_reference():
    """`run_offline` hands over float32 time grids; the reference sorts the (K, 3) array np.concatenate built from them, i.e.
    float64 keys, and NumPy's unstable sort orders ties differently for 32- and 64-bit keys once a clip has more than 16 notes.  Chord-heavy
    maps (every onset lands on one of a few frames, so most notes tie), float32 grids: the host decoder against the oracle restatement of
    the reference (pinned to reference-generated fixtures in tests/test_oracle_notes.py), row order included."""
    rng = np.random.default_rng(77)
    for T, sr in ((120, 22050), (625, 16000), (300, 44100)):
        times = (np.arange(T) * 512 / sr).astype(np.float32)
        on = np.zeros((88, T), dtype=np.float32)
        for f in rng.choice(T - 8, 6, replace=False):
            on[rng.choice(88, 25, replace=False), f] = 1               # 25-note chords on six frames
        mp = on.copy()
        for k, f in zip(*np.nonzero(on)):
            mp[k, f:f + rng.integers(1, 8)] = 1
        ref = notes_np.note_transcriber(mp, on, times)
        got = multi_pitch_to_notes(mp, times, 21, on)
        assert ref.shape[0] > 100 and got.dtype == np.float64
        assert np.array_equal(got, ref), (T, sr)


def test_pitch_list_wrapper():
    mp = np.zeros((88, 4), dtype=np.float32)
    mp[[3, 10], 1] = 1
    mp[87, 3] = 1
    times, pl = PitchListWrapper(tools.PianoProfile()).estimate({tools.KEY_MULTIPITCH: mp, tools.KEY_TIMES: np.arange(4.0)})
    assert [list(p) for p in pl] == [[], [24.0, 31.0], [], [108.0]]


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_device_decoder_bit_exact_with_reference(name):
    torch = pytest.importorskip('torch')
    from amt_tools_amd.transcribe import decode_notes_batch
    g = load_golden(name)
    mp = torch.from_numpy(g['multi_pitch'])[None].cuda()
    on = torch.from_numpy(g['onsets'])[None].cuda() if int(g['with_onsets']) else None
    notes = decode_notes_batch(on, mp, g['times'])[0]
    assert notes.shape == g['notes'].shape and np.array_equal(notes, g['notes'])


@pytest.mark.gpu
def test_device_decoder_full_size_batch_vs_host():
    torch = pytest.importorskip('torch')
    from amt_tools_amd.transcribe import decode_notes_batch
    rng = np.random.default_rng(1)
    B, T = 6, 625
    mp = (rng.random((B, 88, T)) < 0.2).astype(np.float32)
    on = (rng.random((B, 88, T)) < 0.05).astype(np.float32)
    mp[0] = 1.0                                                   # every cell active: one long note per key
    on[1] = 0.0
    on[2, :, ::2] = 1.0                                           # the densest possible impulse train
    on[2, :, 1::2] = 0.0
    times = np.arange(T) * 512 / 22050.0
    got = decode_notes_batch(torch.from_numpy(on).cuda(), torch.from_numpy(mp).cuda(), times)
    for b in range(B):
        assert np.array_equal(got[b], multi_pitch_to_notes(mp[b], times, 21, on[b]))
    got2 = decode_notes_batch(None, torch.from_numpy(mp).cuda(), times)
    for b in range(B):
        assert np.array_equal(got2[b], multi_pitch_to_notes(mp[b], times, 21, None))


@pytest.mark.gpu
def test_device_decoder_per_clip_grids_many_clips_and_capacity_retry():
    """amtx_notes_rows: per-clip time grids ((B,T), float32 as run_offline hands them over), more clips than the scan kernel's 1024-clip
    chunk, empty clips in between, and a first buffer that is too small (the retry with the exact size must return the same notes)."""
    torch = pytest.importorskip('torch')
    from amt_tools_amd.transcribe import decode_notes_batch, decode_notes_batch_async
    rng = np.random.default_rng(5)
    B, T = 1500, 96
    mp = (rng.random((B, 88, T)) < 0.08).astype(np.float32)
    on = (rng.random((B, 88, T)) < 0.02).astype(np.float32)
    mp[::7] = 0.0
    on[::7] = 0.0                                                 # clips without a single note
    hops = rng.integers(128, 1024, B)
    times = (np.arange(T)[None, :] * hops[:, None] / 22050.0).astype(np.float32)
    mpd, ond = torch.from_numpy(mp).cuda(), torch.from_numpy(on).cuda()
    got = decode_notes_batch(ond, mpd, times)
    small = decode_notes_batch_async(ond, mpd, times, rows_capacity=100).result()
    assert len(got) == B
    for b in range(B):
        ref = multi_pitch_to_notes(mp[b], times[b], 21, on[b])
        assert got[b].dtype == np.float64 and got[b].shape == ref.shape and np.array_equal(got[b], ref), b
        assert np.array_equal(small[b], ref), b
    assert got[0].shape == (0, 3)


def test_order_pool_returns_the_in_process_order(monkeypatch):
    """amt_tools_amd/_order_pool.py: the worker processes only repeat NumPy's argsorts on their share of the clips -- same rows, same order
    as the in-process loop, empty clips included; a pool that cannot work falls back to the in-process loop."""
    from amt_tools_amd import _order_pool as op
    rng = np.random.default_rng(3)
    B = 200
    ns = rng.integers(0, 60, B)
    ns[::9] = 0
    off = np.concatenate([[0], np.cumsum(ns)])
    E = int(off[-1])
    rows = np.empty((E, 3))
    rows[:, 0] = rng.integers(0, 12, E) * 0.0232                   # many equal onsets: the order among them is the point
    rows[:, 1] = rows[:, 0] + rng.random(E)
    rows[:, 2] = rng.integers(21, 109, E)
    onset = np.ascontiguousarray(rows[:, 0])
    monkeypatch.setenv('AMTX_NOTE_WORKERS', '0')
    ref = op.order_batch(rows, onset, off, B)
    for b in range(B):
        lo, hi = off[b], off[b + 1]
        assert ref[b].shape == (hi - lo, 3)
        if hi > lo:
            assert np.array_equal(ref[b], rows[lo:hi][op.reference_order(onset[lo:hi])])
    monkeypatch.setenv('AMTX_NOTE_WORKERS', '3')
    got = op.order_batch(rows, onset, off, B, min_clips=1)
    assert op._POOL is not None and not op._POOL.failed and len(op._POOL.procs) == 3
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    # a dead worker: the batch is still ordered (in-process), and the pool stays off
    op._POOL.procs[1].kill()
    op._POOL.procs[1].wait()
    got = op.order_batch(rows, onset, off, B, min_clips=1)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert op._POOL.failed
    op._POOL = None
