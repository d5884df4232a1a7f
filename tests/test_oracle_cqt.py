"""Known-answer tests pinning the CQT/VQT/HCQT restatement (oracle/cqt_np.py).  Parity against librosa itself is
UNPINNED (not installed; the inter-octave decimation filter is implementation-defined there anyway)."""
import numpy as np
import pytest

from oracle import cqt_np as cq


def _tone(freq, n=40000, sr=22050, amp=0.8):
    return amp * np.cos(2 * np.pi * freq * np.arange(n) / sr + 0.4)


@pytest.mark.parametrize('lv', ['0.10', '0.9'])
@pytest.mark.parametrize('k', [5, 30, 47, 70])
def test_sinusoid_at_bin_centre_peaks_with_analytic_magnitude(k, lv):
    sr, bpo, n_bins, amp = 22050, 12, 72, 0.8
    freqs = cq.C1_HZ * 2.0 ** (np.arange(n_bins) / bpo)
    V = np.abs(cq.vqt(_tone(freqs[k], amp=amp), sr, 512, None, n_bins, bpo, 0.0, lv))
    mid = V[:, V.shape[1] // 2]
    assert np.argmax(mid) == k
    L = (1.0 / cq.alpha_of(bpo, lv)) * sr / freqs[k]
    # steady-state response of the L1-normalised, length-scaled wavelet to a centred sinusoid: (A/2) sqrt(L)
    assert abs(mid[k] / (0.5 * amp * np.sqrt(L)) - 1.0) < 0.03
    assert mid[(k + 6) % n_bins] < 0.02 * mid[k]            # half an octave away: > 34 dB down


def test_linearity_and_frame_counts():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(30000)
    a = cq.vqt(y, 22050, 512, None, 48, 12, 0.0)
    b = cq.vqt(3.0 * y, 22050, 512, None, 48, 12, 0.0)
    np.testing.assert_allclose(b, 3.0 * a, rtol=1e-9, atol=1e-12)
    assert a.shape == (48, 1 + 30000 // 512)


def test_decimator_is_a_unit_gain_lowpass_with_sqrt2_scaling():
    h = cq.decimation_filter()
    assert len(h) == 301 and abs(h.sum() - 1) < 1e-12 and np.allclose(h, h[::-1])
    H = np.abs(np.fft.rfft(h, 8192))
    f = np.arange(len(H)) / 8192.0
    assert np.all(np.abs(H[f <= 0.22] - 1) < 1e-3) and np.all(H[f >= 0.26] < 1e-4)
    y = np.ones(1001)
    d = cq.decimate2(y)
    assert len(d) == 501 and abs(d[250] - np.sqrt(2)) < 1e-9


def test_early_downsample_counts_of_the_hcqt_config():
    """HCQT(n_bins=72, 12/oct, sr 22050): harmonics 0.5, 1 are decimated early (2x, 1x), h >= 2 are not."""
    counts = []
    for h in [0.5, 1, 2, 3, 4, 5]:
        freqs = h * cq.C1_HZ * 2.0 ** (np.arange(72) / 12)
        _, cutoff = cq.wavelet_lengths(freqs, 22050, 0.0, cq.alpha_of(12))
        counts.append(cq.early_downsample_count(11025.0, cutoff, 512, 6))
    assert counts == [2, 1, 0, 0, 0, 0]
    with pytest.raises(ValueError):                              # finding F10: 84 bins cannot run at 22.05 kHz for h = 3
        cq.vqt(np.zeros(4096), 22050, 512, 3 * cq.C1_HZ, 84, 12, 0.0)


def test_feature_wrappers_shapes_and_ranges():
    y = _tone(440.0, n=20000) + 0.1 * _tone(1234.0, n=20000)
    f = cq.cqt_process_audio(y, sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    assert f.shape == (1, 72, 1 + 20000 // 512) and f.max() == 1.0 and f.min() >= 0.0
    v = cq.vqt_process_audio(y, n_bins=48)                       # default gamma > 0
    assert v.shape == (1, 48, 40)
    hc = cq.hcqt_process_audio(y, n_bins=72)
    T = min(cq.vqt_expected_frames(20000, 22050, 512, h * cq.C1_HZ, 72, 12, 0.0) for h in [0.5, 1, 2, 3, 4, 5])
    assert hc.shape == (6, 72, T)
    for c in range(6):
        assert hc[c].max() <= 1.0 and hc[c].min() >= 0.0
