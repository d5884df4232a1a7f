"""The note-decoding oracle (oracle/notes_np.py) bit-exact against the reference NoteTranscriber's
output recorded in tests/golden/notes_*.npz."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import notes_np


@pytest.mark.parametrize('name', ['notes_dense.npz', 'notes_sparse.npz', 'notes_noonsets.npz', 'notes_empty.npz', 'notes_f32times.npz'])
def test_notes_bit_exact(name):
    g = load_golden(name)
    onsets = g['onsets'] if int(g['with_onsets']) else None
    notes = notes_np.note_transcriber(g['multi_pitch'], onsets, g['times'])
    assert notes.shape == g['notes'].shape
    assert np.array_equal(notes, g['notes'])   # bit-exact float64 times and pitches, same row order


def test_hop_estimate_matches_uniform_grid():
    times = np.arange(100) * 512 / 22050.0
    assert notes_np.estimate_hop_length(times) == np.median(np.diff(times))
    with pytest.raises(ValueError):
        notes_np.estimate_hop_length(np.array([]))
