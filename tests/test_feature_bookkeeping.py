"""FeatureModule bookkeeping (row A0): frame counts, sample ranges, required samples, sizes, names, frame times and the dB
post-processing of the mirror classes against vectors recorded from the REAL reference's STFT / MelSpec classes
(tests/golden/feature_bookkeeping.npz, tools/gen_golden.py).  No GPU: none of these methods touch the device."""
import numpy as np
import pytest

from conftest import load_golden
from amt_tools_amd.features import STFT, MelSpec


def _mods():
    return {'stft_c': STFT(sample_rate=22050, hop_length=512, n_fft=2048),
            'stft_nc': STFT(sample_rate=16000, hop_length=256, n_fft=1024, win_length=800, center=False),
            'mel_c': MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048),
            'mel_nc': MelSpec(sample_rate=16000, hop_length=512, n_mels=64, n_fft=2048, center=False)}


@pytest.mark.parametrize('key', ['stft_c', 'stft_nc', 'mel_c', 'mel_nc'])
def test_bookkeeping_matches_reference(key):
    g = load_golden('feature_bookkeeping.npz')
    m = _mods()[key]
    got = np.array([m.get_expected_frames(np.zeros(int(n))) for n in g['lengths']])
    np.testing.assert_array_equal(got, g[key + '_expected_frames'])
    for f, lo, hi, ln in zip(g['frames'], g[key + '_sample_range_min'], g[key + '_sample_range_max'], g[key + '_sample_range_len']):
        r = m.get_sample_range(int(f))
        assert (int(np.min(r)), int(np.max(r)), len(r)) == (int(lo), int(hi), int(ln)), (key, int(f))
    assert m.get_num_samples_required() == int(g[key + '_num_samples_required'])
    assert m.get_feature_size() == int(g[key + '_feature_size'])
    assert m.get_num_channels() == int(g[key + '_num_channels'])
    assert m.features_name() == str(g[key + '_name'])
    np.testing.assert_allclose(m.get_times(np.zeros(5000)), g[key + '_times'], rtol=0, atol=1e-12)
    assert m.get_sample_rate() in (16000, 22050) and m.get_hop_length() in (256, 512)


def test_db_post_processing_matches_reference_formulas():
    g = load_golden('feature_bookkeeping.npz')
    mods = _mods()
    np.testing.assert_allclose(mods['stft_c'].post_proc(g['db_in'].copy()), g['stft_post'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(mods['mel_c'].post_proc(g['db_in'].copy()), g['mel_post'], rtol=0, atol=1e-6)
    assert g['mel_post'].shape == (1, 5, 12) and g['mel_post'].max() == 1.0
