"""
ORACLE (test infrastructure only) -- NumPy restatement of the CQT / VQT / HCQT front-end of amt-tools.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

What it restates
----------------
* amt_tools/features/vqt.py:21-62,167-195   VQT.process_audio: librosa.vqt -> abs -> amplitude_to_db(ref=max) -> /80+1
* amt_tools/features/cqt.py:22               CQT = VQT with gamma = 0
* amt_tools/features/hvqt.py:16-58,107-133   HVQT/HCQT: one VQT per harmonic h*fmin, each dB-normalised on its own,
                                             truncated to the minimum frame count, stacked on the channel axis
* amt_tools/features/vqt.py:64-165           frame / sample bookkeeping (early-downsample count)

The transform itself is third-party **librosa** (requirements.txt:3 `librosa>=0.9.1`, unpinned, not vendored, not
installed here): `librosa.vqt`, `filters.wavelet`, `filters.wavelet_lengths`, `util.sparsify_rows`,
`core.constantq.__early_downsample_count`, restated from the published algorithm in its
own frequency-domain form: per octave, a "ones"-window STFT of the progressively decimated signal times a sparse
FFT-domain wavelet basis.  `lv` selects the librosa convention: '0.10' -> alpha = (r^2-1)/(r^2+1), zero centre
padding; '0.9' -> alpha = r-1, reflect padding.

PARITY UNPINNED vs librosa.  In addition the 2:1 decimation low-pass between octaves is implementation-defined in
librosa (soxr / resampy internals); this project uses its OWN documented filter (`decimation_filter`):
a 301-tap Kaiser(beta=10) windowed sinc with -6 dB point at 0.239 fs (soxr-HQ-like: passband to ~0.91 of the new
Nyquist), zero-phase, output length ceil(n/2), scaled by sqrt(2) (librosa `resample(..., scale=True)`).  Pinned by
known-answer tests (tests/test_oracle_cqt.py): a stationary sinusoid at a bin centre peaks in that bin with the
analytic magnitude, octave-shift invariance, bookkeeping identities.
"""

import numpy as np

from .frontend_np import amplitude_to_db, post_proc

C1_HZ = 32.70319566257483        # librosa.note_to_hz('C1')
HANN_BANDWIDTH = 1.50018310546875  # librosa.filters.window_bandwidth('hann')
DECIM_HALF = 150                 # decimator taps = 2 * DECIM_HALF + 1
DECIM_CUTOFF = 0.239             # cycles/sample at the input rate
DECIM_BETA = 10.0


def decimation_filter():
    n = np.arange(-DECIM_HALF, DECIM_HALF + 1, dtype=np.float64)
    h = 2.0 * DECIM_CUTOFF * np.sinc(2.0 * DECIM_CUTOFF * n) * np.kaiser(2 * DECIM_HALF + 1, DECIM_BETA)
    return h / h.sum()


def decimate2(y):
    """y[n] -> sqrt(2) * sum_k h[k] y[2n + k - DECIM_HALF], zero beyond the ends, length ceil(len/2)."""
    h = decimation_filter()
    n_out = (len(y) + 1) // 2
    full = np.convolve(np.asarray(y, dtype=np.float64), h[::-1], mode='full')      # correlation with symmetric h
    # full[m] = sum_k y[m - (N-1) + k'] ... with symmetric h: out[n] = full[2n + DECIM_HALF]
    return np.sqrt(2.0) * full[DECIM_HALF:DECIM_HALF + 2 * n_out:2]


def alpha_of(bins_per_octave, lv='0.10'):
    r = 2.0 ** (1.0 / bins_per_octave)
    return (r - 1.0) if lv == '0.9' else (r * r - 1.0) / (r * r + 1.0)


def wavelet_lengths(freqs, sr, gamma, alpha, filter_scale=1.0):
    Q = float(filter_scale) / alpha
    cutoff = np.max(freqs * (1 + 0.5 * HANN_BANDWIDTH / Q) + 0.5 * gamma)
    lengths = Q * sr / (freqs + gamma / alpha)
    return lengths, cutoff


def num_two_factors(x):
    if x <= 0:
        return 0
    n = 0
    while x % 2 == 0:
        n += 1
        x //= 2
    return n


def early_downsample_count(nyquist, filter_cutoff, hop_length, n_octaves):
    c1 = max(0, int(np.ceil(np.log2(nyquist / filter_cutoff)) - 1) - 1)
    c2 = max(0, num_two_factors(hop_length) - n_octaves + 1)
    return min(c1, c2)


def wavelet_basis(freqs, sr, gamma, alpha):
    """filters.wavelet(..., norm=1, pad_fft=True, window='hann'): complex (n_filters, n_fft), lengths."""
    lengths, _ = wavelet_lengths(freqs, sr, gamma, alpha)
    filts = []
    for ilen, freq in zip(lengths, freqs):
        t = np.arange(-ilen // 2, ilen // 2, dtype=float)
        sig = np.exp(1j * t * 2 * np.pi * freq / sr)
        n = len(sig)
        sig = sig * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))
        sig = sig / np.sum(np.abs(sig))
        filts.append(sig)
    n_fft = int(2.0 ** (np.ceil(np.log2(max(lengths)))))
    out = np.zeros((len(freqs), n_fft), dtype=np.complex128)
    for i, sig in enumerate(filts):
        lpad = (n_fft - len(sig)) // 2
        out[i, lpad:lpad + len(sig)] = sig
    return out, lengths, n_fft


def sparsify_rows(x, quantile=0.01):
    mags = np.abs(x)
    norms = np.sum(mags, axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cumulative = np.cumsum(mag_sort / norms, axis=1)
    idx = np.argmin(cumulative < quantile, axis=1)
    out = np.zeros_like(x)
    for i, j in enumerate(idx):
        keep = mags[i] >= mag_sort[i, j]
        out[i, keep] = x[i, keep]
    return out


def stft_ones(y, n_fft, hop, lv):
    mode = 'reflect' if lv == '0.9' else 'constant'
    yp = np.pad(y, (n_fft // 2, n_fft // 2), mode=mode)
    T = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(T)[None, :]
    return np.fft.rfft(yp[idx], axis=0)


def vqt(y, sr=22050, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, gamma=0.0, lv='0.10'):
    """librosa.vqt restated (scale=True, norm=1, sparsity=0.01, filter_scale=1, tuning=0).  Complex (n_bins, T)."""
    y = np.asarray(y, dtype=np.float64)
    if fmin is None:
        fmin = C1_HZ
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    alpha = alpha_of(bins_per_octave, lv)
    lengths, cutoff = wavelet_lengths(freqs, sr, gamma, alpha)
    nyquist = sr / 2.0
    if cutoff > nyquist:
        raise ValueError(f'Wavelet basis with max frequency={np.max(freqs)} would exceed the Nyquist frequency={nyquist}.')
    count = early_downsample_count(nyquist, cutoff, hop_length, n_octaves)
    for _ in range(count):
        y = decimate2(y)
    sr = sr / 2.0 ** count
    hop_length //= 2 ** count
    resp = []
    my_y, my_sr, my_hop = y, sr, hop_length
    for i in range(n_octaves):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        basis, blen, n_fft = wavelet_basis(freqs[sl], my_sr, gamma, alpha)
        basis = basis * (blen[:, None] / float(n_fft))
        fft_basis = sparsify_rows(np.fft.fft(basis, n=n_fft, axis=1)[:, :n_fft // 2 + 1])
        fft_basis = fft_basis * np.sqrt(sr / my_sr)
        resp.append(fft_basis @ stft_ones(my_y, n_fft, my_hop, lv))
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            my_y = decimate2(my_y)
    max_col = min(r.shape[-1] for r in resp)
    out = np.empty((n_bins, max_col), dtype=np.complex128)
    end = n_bins
    for r in resp:
        n_oct = r.shape[0]
        if end < n_oct:
            out[:end] = r[-end:, :max_col]
        else:
            out[end - n_oct:end] = r[:, :max_col]
        end -= n_oct
    lengths, _ = wavelet_lengths(freqs, sr, gamma, alpha)
    return out / np.sqrt(lengths[:, None])


def default_gamma(bins_per_octave):
    """VQT.__init__ (features/vqt.py:51-58): alpha kept in the librosa-0.9 convention, gamma = 24.7 alpha / 0.108."""
    return 24.7 * (2.0 ** (1.0 / bins_per_octave) - 1) / 0.108


def vqt_process_audio(audio, sample_rate=22050, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, gamma=None,
                      decibels=True, lv='0.10'):
    """VQT.process_audio (features/vqt.py:167-195): (1, n_bins, T)."""
    if gamma is None:
        gamma = default_gamma(bins_per_octave)
    mag = np.abs(vqt(audio, sample_rate, hop_length, fmin, n_bins, bins_per_octave, gamma, lv))
    if decibels:
        return post_proc(amplitude_to_db(mag))
    return np.expand_dims(mag, axis=0)


def cqt_process_audio(audio, **kw):
    kw['gamma'] = 0.0
    return vqt_process_audio(audio, **kw)


def vqt_early_ds_count(sample_rate, hop_length, fmin, n_bins, bins_per_octave, gamma):
    """VQT.get_early_ds_count (features/vqt.py:64-100): the reference's own estimate (librosa-0.9 Q)."""
    fmax = fmin * 2.0 ** ((n_bins - 1) / bins_per_octave)
    cQ = 1.0 / (2.0 ** (1. / bins_per_octave) - 1)
    freq_cutoff = fmax * (1 + 0.5 * HANN_BANDWIDTH / cQ) + 0.5 * gamma
    n_octs = int(np.ceil(float(n_bins) / bins_per_octave))
    return early_downsample_count(sample_rate / 2.0, freq_cutoff, hop_length, n_octs)


def vqt_expected_frames(num_samples, sample_rate, hop_length, fmin, n_bins, bins_per_octave, gamma):
    """VQT.get_expected_frames (features/vqt.py:102-134)."""
    early = vqt_early_ds_count(sample_rate, hop_length, fmin, n_bins, bins_per_octave, gamma)
    n_octs = int(np.ceil(float(n_bins) / bins_per_octave))
    k = np.arange(early, early + n_octs)
    sig_lens = np.ceil(num_samples / (2 ** k))
    hop_lens = hop_length // (2 ** k)
    return int(min(sig_lens // hop_lens + 1))


def hvqt_process_audio(audio, sample_rate=22050, hop_length=512, fmin=None, harmonics=None, n_bins=84, bins_per_octave=12,
                       gamma=None, decibels=True, lv='0.10'):
    """HVQT.process_audio (features/hvqt.py:107-133): (H, n_bins, T)."""
    if fmin is None:
        fmin = C1_HZ
    if harmonics is None:
        harmonics = [0.5, 1, 2, 3, 4, 5]
    harmonics = sorted(harmonics)
    g = default_gamma(bins_per_octave) if gamma is None else gamma
    num_frames = min(vqt_expected_frames(len(audio), sample_rate, hop_length, h * fmin, n_bins, bins_per_octave, g) for h in harmonics)
    feats = [vqt_process_audio(audio, sample_rate, hop_length, h * fmin, n_bins, bins_per_octave, g, decibels, lv)[..., :num_frames]
             for h in harmonics]
    return np.concatenate(feats, axis=0)


def hcqt_process_audio(audio, **kw):
    kw['gamma'] = 0.0
    return hvqt_process_audio(audio, **kw)
