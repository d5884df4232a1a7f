"""
Deterministic synthetic inputs for tests, golden generation and bench.py.

* `synth_clip(i, ...)`      -- the synthetic 22.05 kHz clip of SURVEY.md section 8(d): eight decaying
                               harmonic notes + low-level noise, RMS-normalised the way
                               amt_tools/tools/utils.py:2789-2814 (rms_norm) does.
* `of_state_dict_shapes()`  -- parameter/buffer names and shapes of the reference's
                               OnsetsFrames / OnsetsFrames2 `state_dict()`.
* `synth_state_dict(...)`   -- seed -> weights, NumPy PCG64 only (stable across NumPy versions), so a
                               golden fixture needs to store the seed, not 19 MB of weights.
"""

from collections import OrderedDict

import numpy as np

__all__ = ['synth_clip', 'synth_batch', 'synth_labels', 'of_state_dict_shapes', 'synth_state_dict', 'tabcnn_state_dict_shapes',
           'synth_tabcnn_state_dict',
           'CLIP_SAMPLES', 'CLIP_FRAMES']

CLIP_SAMPLES = 319999   # max(get_sample_range(625)) -- amt_tools/datasets/common.py:116
CLIP_FRAMES = 625       # examples/papers/of_1.py:44


def synth_clip(i, num_samples=CLIP_SAMPLES, sample_rate=22050, num_notes=8):
    """One synthetic clip, float32 (N,), RMS-normalised."""
    rng = np.random.default_rng(1234 + int(i))
    t = np.arange(num_samples, dtype=np.float64) / sample_rate
    dur_total = num_samples / sample_rate
    y = np.zeros(num_samples, dtype=np.float64)
    for _ in range(num_notes):
        midi = int(rng.integers(21, 109))
        onset = rng.uniform(0.0, max(1e-3, min(12.0, dur_total * 0.83)))
        dur = rng.uniform(0.2, 2.0)
        amp = rng.uniform(0.1, 1.0)
        f0 = 440.0 * 2.0 ** ((midi - 69) / 12.0)
        lo = int(onset * sample_rate)
        hi = min(num_samples, int((onset + dur) * sample_rate))
        if hi <= lo:
            continue
        tt = t[lo:hi]
        env = np.exp(-3.0 * (tt - onset))
        note = np.zeros(hi - lo)
        for k in range(1, 7):
            if k * f0 < sample_rate / 2:
                note += (1.0 / k) * np.sin(2 * np.pi * k * f0 * tt + rng.uniform(0, 2 * np.pi))
        y[lo:hi] += amp * env * note
    y += rng.normal(0.0, 1e-3, num_samples)
    rms = np.sqrt(np.mean(y ** 2))
    if rms > 0:
        y = y / rms
    return y.astype(np.float32)


def synth_batch(first, count, num_samples=CLIP_SAMPLES, sample_rate=22050):
    return np.stack([synth_clip(first + j, num_samples, sample_rate) for j in range(count)])


def synth_labels(i, num_frames=CLIP_FRAMES, num_keys=88):
    """Bernoulli label maps for the training config (seed 4321 + i): (multi_pitch, onsets) float32 (88,T)."""
    rng = np.random.default_rng(4321 + int(i))
    mp = (rng.random((num_keys, num_frames)) < 0.05).astype(np.float32)
    on = (rng.random((num_keys, num_frames)) < 0.01).astype(np.float32)
    return mp, on


def _acoustic_shapes(prefix, dim_in, dim_out, in_channels, mc):
    nf1, nf2, nf3 = 16 * mc, 16 * mc, 32 * mc
    s = OrderedDict()
    for name, cin, cout in (('layer1', in_channels, nf1), ('layer2', nf1, nf2), ('layer3', nf2, nf3)):
        s[f'{prefix}.{name}.0.weight'] = (cout, cin, 3, 3)
        s[f'{prefix}.{name}.0.bias'] = (cout,)
        s[f'{prefix}.{name}.1.weight'] = (cout,)
        s[f'{prefix}.{name}.1.bias'] = (cout,)
        s[f'{prefix}.{name}.1.running_mean'] = (cout,)
        s[f'{prefix}.{name}.1.running_var'] = (cout,)
        s[f'{prefix}.{name}.1.num_batches_tracked'] = ()
    s[f'{prefix}.fc1.0.weight'] = (dim_out, nf3 * (dim_in // 4))
    s[f'{prefix}.fc1.0.bias'] = (dim_out,)
    return s


def _lstm_shapes(prefix, dim_in, dim_out):
    h = dim_out // 2
    s = OrderedDict()
    for sfx in ('', '_reverse'):
        s[f'{prefix}.mlm.weight_ih_l0{sfx}'] = (4 * h, dim_in)
        s[f'{prefix}.mlm.weight_hh_l0{sfx}'] = (4 * h, h)
        s[f'{prefix}.mlm.bias_ih_l0{sfx}'] = (4 * h,)
        s[f'{prefix}.mlm.bias_hh_l0{sfx}'] = (4 * h,)
    return s


def _bank_shapes(prefix, dim_in, dim_out):
    return OrderedDict([(f'{prefix}.output_layer.weight', (dim_out, dim_in)),
                        (f'{prefix}.output_layer.bias', (dim_out,))])


def of_state_dict_shapes(dim_in=229, in_channels=1, model_complexity=2, dim_out=88, offsets=False):
    """Names/shapes of OnsetsFrames(2).state_dict() -- amt_tools/models/onsetsframes.py:22-65,199-233."""
    dim_am = 256 * model_complexity
    dim_lm = 256 * (model_complexity - 1)
    s = OrderedDict()
    s.update(_acoustic_shapes('onset_head.0', dim_in, dim_am, in_channels, model_complexity))
    s.update(_lstm_shapes('onset_head.1', dim_am, dim_lm))
    s.update(_bank_shapes('onset_head.2', dim_lm, dim_out))
    s.update(_acoustic_shapes('pitch_head.0', dim_in, dim_am, in_channels, model_complexity))
    s.update(_bank_shapes('pitch_head.1', dim_am, dim_out))
    dim_aj = (3 if offsets else 2) * dim_out
    s.update(_lstm_shapes('adjoin.0', dim_aj, dim_lm))
    s.update(_bank_shapes('adjoin.1', dim_lm, dim_out))
    if offsets:
        s.update(_acoustic_shapes('offset_head.0', dim_in, dim_am, in_channels, model_complexity))
        s.update(_lstm_shapes('offset_head.1', dim_am, dim_lm))
        s.update(_bank_shapes('offset_head.2', dim_lm, dim_out))
    return s


def synth_state_dict(seed=0, **kwargs):
    """Deterministic weights for every key of `of_state_dict_shapes(**kwargs)` (NumPy arrays, float32;
    num_batches_tracked int64).  Keys are visited in sorted order so the stream is layout-independent."""
    shapes = of_state_dict_shapes(**kwargs)
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key in sorted(shapes):
        shape = shapes[key]
        if key.endswith('num_batches_tracked'):
            val = np.array(7, dtype=np.int64)
        elif key.endswith('running_mean'):
            val = rng.normal(0.0, 0.1, shape).astype(np.float32)
        elif key.endswith('running_var'):
            val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif '.1.weight' in key and len(shape) == 1:       # BatchNorm gamma
            val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1:                                # biases / BatchNorm beta
            val = rng.normal(0.0, 0.05, shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(3.0 / fan_in)                    # unit-gain uniform
            if key.endswith('output_layer.weight'):
                bound *= 4.0                                 # spread the logits away from 0
            val = rng.uniform(-bound, bound, shape).astype(np.float32)
        sd[key] = val
    return OrderedDict((k, sd[k]) for k in shapes)


def tabcnn_state_dict_shapes(dim_in=192, in_channels=1, model_complexity=1, num_groups=6, num_classes=21):
    """Names/shapes of TabCNN.state_dict() -- amt_tools/models/tabcnn.py:47-87 (frame width 9)."""
    nf1, nf2 = 32 * model_complexity, 64 * model_complexity
    emb = nf2 * ((dim_in - 6) // 2) * ((9 - 6) // 2)
    fc = 128 * model_complexity
    return OrderedDict([('conv.0.weight', (nf1, in_channels, 3, 3)), ('conv.0.bias', (nf1,)),
                        ('conv.2.weight', (nf2, nf1, 3, 3)), ('conv.2.bias', (nf2,)),
                        ('conv.4.weight', (nf2, nf2, 3, 3)), ('conv.4.bias', (nf2,)),
                        ('dense.0.weight', (fc, emb)), ('dense.0.bias', (fc,)),
                        ('dense.3.output_layer.weight', (num_groups * num_classes, fc)),
                        ('dense.3.output_layer.bias', (num_groups * num_classes,))])


def synth_tabcnn_state_dict(seed=0, **kwargs):
    """Deterministic TabCNN weights (same recipe as synth_state_dict)."""
    shapes = tabcnn_state_dict_shapes(**kwargs)
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key in sorted(shapes):
        shape = shapes[key]
        if len(shape) == 1:
            val = rng.normal(0.0, 0.05, shape).astype(np.float32)
        else:
            bound = np.sqrt(3.0 / int(np.prod(shape[1:]))) * (4.0 if key.endswith('output_layer.weight') else 1.0)
            val = rng.uniform(-bound, bound, shape).astype(np.float32)
        sd[key] = val
    return OrderedDict((k, sd[k]) for k in shapes)
