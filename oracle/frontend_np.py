"""
ORACLE (test infrastructure only) -- CPU restatement of the spectral front-end
of amt-tools' hot path: STFT / MelSpec `process_audio` + dB post-processing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (amt_tools_amd.*) never does.

What it restates
----------------
* amt_tools/features/stft.py:42-77      STFT.process_audio        (|stft| -> post_proc)
* amt_tools/features/mel.py:40-96       MelSpec.process_audio     (melspectrogram -> power_to_db)
* amt_tools/features/common.py:181-230  to_decibels / post_proc   (dB -> /80 -> +1 -> channel axis)
* amt_tools/features/waveform.py:43-153 frame bookkeeping

The arithmetic itself lives in the third-party dependency **librosa**
(requirements.txt:3, `librosa>=0.9.1`, no upper pin, not vendored, NOT installed
in this image).  The transform is therefore restated from librosa's published
algorithm with the librosa version as an explicit
parameter `lv` ('0.10' -> zero centre padding, '0.9' -> reflect padding).

PARITY UNPINNED vs librosa: there is no runnable librosa here and the reference
holds no golden vectors for this boundary.  The restatement is pinned instead by
oracle-free known-answer tests (tests/test_oracle_frontend.py): analytic sinusoid
magnitudes, Parseval, mel triangle geometry / Slaney area normalisation, dB
range end points and frame-count identities, and cross-checked against two
independent implementations present in the image: scipy.signal.ShortTimeFFT (framed
FFT) and transformers.audio_utils (HF's librosa-compatible mel filterbank and log-mel
chain): agreement to 1e-9 / float32 rounding / 1e-5 on the scaled features.
The only librosa OUTPUTS at hand -- the example values its docstrings print for hz_to_mel,
mel_to_hz, mel_frequencies(n_mels=40) and filters.mel(sr=22050, n_fft=2048) -- are reproduced at
their printed precision (test_mel_scale_and_filterbank_reproduce_the_values_librosa_documents).
"""

import numpy as np

__all__ = [
    'hann_periodic', 'fft_window', 'pad_center_audio', 'stft', 'hz_to_mel', 'mel_to_hz',
    'mel_frequencies', 'mel_filterbank', 'melspectrogram', 'power_to_db',
    'amplitude_to_db', 'post_proc', 'melspec_process_audio', 'stft_process_audio',
    'expected_frames', 'sample_range', 'frame_times',
]


def hann_periodic(win_length, dtype=np.float64):
    """scipy.signal.get_window('hann', win_length, fftbins=True): 0.5 - 0.5 cos(2 pi n / win)."""
    n = np.arange(win_length, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(dtype)


def fft_window(n_fft, win_length=None):
    """Periodic Hann of win_length, zero padded symmetrically to n_fft (librosa.util.pad_center)."""
    if win_length is None:
        win_length = n_fft
    w = hann_periodic(win_length)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft, dtype=np.float64)
    out[lpad:lpad + win_length] = w
    return out


def pad_center_audio(y, n_fft, lv='0.10'):
    """librosa.stft(center=True) padding: n_fft//2 both sides; zeros (>=0.10) or reflect (0.9)."""
    mode = 'constant' if lv != '0.9' else 'reflect'
    return np.pad(y, (n_fft // 2, n_fft // 2), mode=mode)


def stft(y, n_fft=2048, hop_length=512, win_length=None, center=True, lv='0.10', dtype=np.float64):
    """
    librosa.stft restated.  Returns (1 + n_fft//2, T) complex.
    With dtype=float64 this is the high-precision oracle; librosa itself multiplies the
    float64 window into the float32 frames, transforms in float64 and rounds to complex64.
    """
    y = np.asarray(y, dtype=dtype)
    w = fft_window(n_fft, win_length).astype(dtype)
    if center:
        y = pad_center_audio(y, n_fft, lv)
    if y.shape[-1] < n_fft:
        return np.zeros((1 + n_fft // 2, 0), dtype=np.complex128)
    T = 1 + (y.shape[-1] - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = y[idx] * w[:, None]
    return np.fft.rfft(frames, axis=0)


def hz_to_mel(f, htk=False):
    f = np.asanyarray(f, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz(m, htk=False):
    m = np.asanyarray(m, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_frequencies(n_mels, fmin, fmax, htk=False):
    return mel_to_hz(np.linspace(hz_to_mel(fmin, htk), hz_to_mel(fmax, htk), n_mels), htk)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm='slaney', dtype=np.float32):
    """librosa.filters.mel restated.  (n_mels, 1 + n_fft//2)."""
    if fmax is None:
        fmax = float(sr) / 2
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax, htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    # librosa allocates `weights` in `dtype` (float32), assigns the float64 triangles into it and then
    # scales in place by the float64 area norm: two roundings when dtype is float32.
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=dtype)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == 'slaney':
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, None]
    return weights


def melspectrogram(y, sr, n_mels=229, n_fft=2048, hop_length=512, win_length=None, center=True,
                   htk=False, lv='0.10', dtype=np.float64):
    """librosa.feature.melspectrogram restated: mel_basis @ |stft|**2 (power=2, fmin=0, fmax=sr/2, slaney)."""
    S = np.abs(stft(y, n_fft, hop_length, win_length, center, lv, dtype)) ** 2
    basis = mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2.0, htk, 'slaney', dtype=np.float32).astype(dtype)
    return basis @ S.astype(dtype)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(S, ref=np.max) restated."""
    S = np.asarray(S)
    ref_value = np.max(S) if S.size else 1.0
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin, ref_value))
    if S.size:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def amplitude_to_db(S, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db(S, ref=np.max) = power_to_db(|S|^2, ref=max|S|^2, amin^2)."""
    mag = np.abs(np.asarray(S))
    ref_value = np.max(mag) if mag.size else 1.0
    power = np.square(mag)
    log_spec = 10.0 * np.log10(np.maximum(amin ** 2, power))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin ** 2, ref_value ** 2))
    if mag.size:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def post_proc(feats_db):
    """features/common.py:218-228: dB -> /80 -> +1 -> add channel axis."""
    return np.expand_dims(feats_db / 80 + 1, axis=0)


def melspec_process_audio(audio, sample_rate=16000, hop_length=512, n_mels=229, n_fft=2048,
                          win_length=None, center=True, htk=False, decibels=True, lv='0.10',
                          dtype=np.float64):
    """MelSpec.process_audio (features/mel.py:40-76) restated.  Returns (1, n_mels, T)."""
    audio = np.asarray(audio)
    if audio.shape[-1] == 0:
        return np.zeros((1, n_mels, 0))
    if not center:
        audio = frame_pad(audio, hop_length, win_length if win_length else n_fft, center)
    mel = melspectrogram(audio, sample_rate, n_mels, n_fft, hop_length, win_length, center, htk, lv, dtype)
    if decibels:
        return post_proc(power_to_db(mel))
    return np.expand_dims(mel, axis=0)


def stft_process_audio(audio, hop_length=512, n_fft=2048, win_length=None, center=True,
                       decibels=True, lv='0.10', dtype=np.float64):
    """STFT.process_audio (features/stft.py:42-77) restated.  Returns (1, 1+n_fft//2, T)."""
    audio = np.asarray(audio)
    if audio.shape[-1] == 0:
        # reference quirk (stft.py:57-59): the empty case reports n_fft rows
        return np.zeros((1, n_fft, 0))
    if not center:
        audio = frame_pad(audio, hop_length, win_length if win_length else n_fft, center)
    spec = np.abs(stft(audio, n_fft, hop_length, win_length, center, lv, dtype))
    if decibels:
        return post_proc(amplitude_to_db(spec))
    return np.expand_dims(spec, axis=0)


# ---------------------------------------------------------------- bookkeeping (row A0)

def expected_frames(num_samples, hop_length, win_length=None, center=True):
    """features/common.py:41-66 and features/waveform.py:43-67."""
    if num_samples == 0:
        return 0
    if center:
        return 1 + num_samples // hop_length
    return 1 + ((max(0, (num_samples - win_length)) - 1) // hop_length + 1)


def sample_range(num_frames, hop_length, win_length=None, center=True):
    """features/common.py:68-97 and features/waveform.py:69-96."""
    if num_frames == 0:
        return np.array([0])
    if center:
        max_samples = num_frames * hop_length - 1
        min_samples = max(1, max_samples - hop_length + 1)
        return np.arange(min_samples, max_samples + 1)
    if num_frames == 1:
        return np.arange(1, win_length + 1)
    required = win_length
    return np.arange(1, hop_length + 1) + required + (num_frames - 2) * hop_length


def frame_pad(audio, hop_length, win_length, center):
    """features/common.py:114-166 (divisor_pad / frame_pad)."""
    divisor = sample_range(1, hop_length, win_length, center)[-1]
    if audio.shape[-1] > divisor:
        divisor = hop_length
    pad_amt = divisor - (audio.shape[-1] % divisor)
    if pad_amt > 0 and pad_amt != divisor:
        audio = np.append(audio, np.zeros(pad_amt).astype(np.float32), axis=-1)
    return audio


def frame_times(num_frames, sample_rate, hop_length):
    """librosa.frames_to_time: frames * hop / sr in float64 (features/common.py:232-258)."""
    return (np.arange(num_frames) * hop_length).astype(np.float64) / float(sample_rate)
