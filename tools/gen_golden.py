#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference); the GPU box and
the test-suite never execute it -- they read the small .npz fixtures it wrote to tests/golden/.

It imports the *real* reference classes (amt_tools.models.OnsetsFrames / OnsetsFrames2,
amt_tools.transcribe.NoteTranscriber, amt_tools.tools.*) with import-time stubs for the third-party
packages this image lacks (librosa, mir_eval, jams, ...; recipe from SURVEY.md Appendix C), loads
seed-generated weights (amt_tools_amd.synth.synth_state_dict, so the fixture stores a seed rather
than 19 MB of parameters) and records inputs + the reference's outputs.
"""
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith('__') and n.endswith('__'):
            raise AttributeError(n)
        return MagicMock(name=f'{self.__name__}.{n}')


for m in ['librosa', 'librosa.util', 'librosa.core', 'librosa.core.constantq', 'librosa.filters', 'mir_eval',
          'mir_eval.transcription', 'mir_eval.multipitch', 'mir_eval.util', 'jams', 'mido', 'tensorboardX',
          'mirdata', 'mirdata.datasets', 'sounddevice', 'pynput', 'sacred', 'tqdm']:
    if m in ('tqdm',):
        continue
    s = _Stub(m)
    s.__spec__ = importlib.machinery.ModuleSpec(m, None)
    s.__path__ = []
    sys.modules[m] = s

# two librosa helpers the TabCNN path calls are one-liners whose documented behaviour is restated here so that the
# reference's own code runs (the stub would hand back MagicMocks): symmetric zero padding and note names -> MIDI numbers
def _pad_center(data, size, axis=-1, **kw):
    n = data.shape[axis]
    lpad = (size - n) // 2
    widths = [(0, 0)] * data.ndim
    widths[axis] = (lpad, size - n - lpad)
    return np.pad(data, widths, mode='constant')


def _note_to_midi(note):
    if isinstance(note, (list, tuple)):
        return [_note_to_midi(n) for n in note]
    pc = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}[note[0].upper()]
    rest = note[1:]
    while rest and rest[0] in '#b':
        pc += 1 if rest[0] == '#' else -1
        rest = rest[1:]
    return 12 * (int(rest) + 1) + pc


sys.modules['librosa'].note_to_midi = _note_to_midi
sys.modules['librosa.util'].pad_center = _pad_center
sys.modules['librosa'].util = sys.modules['librosa.util']

import amt_tools                                    # noqa: E402  (the reference)
from amt_tools import tools as rtools               # noqa: E402
from amt_tools.models import OnsetsFrames, OnsetsFrames2, TabCNN   # noqa: E402
from amt_tools.transcribe import NoteTranscriber   # noqa: E402

from amt_tools_amd.synth import synth_state_dict, of_state_dict_shapes, synth_tabcnn_state_dict   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def load_weights(model, seed, **kw):
    sd_np = synth_state_dict(seed, **kw)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd_np.keys()), 'state_dict key order differs from the reference'
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd_np[k].shape), (k, v.shape, sd_np[k].shape)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    return sd_np


def weight_checksum(sd_np):
    return np.array([float(np.abs(np.asarray(v, dtype=np.float64)).sum()) for v in sd_np.values()])


def features(seed, B, C, F, T):
    rng = np.random.default_rng(seed)
    # smooth-ish values in [0, 1] like dB-scaled spectrogram features
    x = rng.random((B, C, F, T)).astype(np.float32)
    x = 0.5 * x + 0.5 * np.roll(x, 1, axis=-1)
    return x.astype(np.float32)


def labels(seed, B, T, p):
    rng = np.random.default_rng(seed)
    return (rng.random((B, 88, T)) < p).astype(np.float32)


def gen_of_eval(name, cls, seed, dim_in, in_channels, mc, B, T, offsets):
    profile = rtools.PianoProfile()
    model = cls(dim_in, profile, in_channels, mc)
    sd_np = load_weights(model, seed, dim_in=dim_in, in_channels=in_channels, model_complexity=mc, offsets=offsets)
    model.eval()
    feats = features(seed + 100, B, in_channels, dim_in, T)
    times = (np.arange(T) * 512 / 22050.0)
    with torch.no_grad():
        batch = {rtools.KEY_FEATS: torch.from_numpy(feats), rtools.KEY_TIMES: torch.from_numpy(np.tile(times, (B, 1)))}
        pre = model.pre_proc(dict(batch))
        raw = model(pre[rtools.KEY_FEATS])
        out = model.run_on_batch(dict(batch))
        pitch_head = model.pitch_head(pre[rtools.KEY_FEATS])
    rec = dict(seed=seed, dim_in=dim_in, in_channels=in_channels, model_complexity=mc, offsets=int(offsets),
               feats=feats, wsum=weight_checksum(sd_np),
               logits_onsets=raw[rtools.KEY_ONSETS].numpy(), logits_multi_pitch=raw[rtools.KEY_MULTIPITCH].numpy(),
               logits_pitch_head=pitch_head.numpy(),
               out_onsets=out[rtools.KEY_ONSETS].numpy(), out_multi_pitch=out[rtools.KEY_MULTIPITCH].numpy(),
               out_times=out[rtools.KEY_TIMES].numpy())
    if offsets:
        rec['logits_offsets'] = raw[rtools.KEY_OFFSETS].numpy()
        rec['out_offsets'] = out[rtools.KEY_OFFSETS].numpy()
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, {k: getattr(v, 'shape', v) for k, v in rec.items()})


# small tensors only: the recurrent matrices of model_complexity 4 are 1536 x 384
MC4_GKEYS = ['onset_head.0.layer1.0.weight', 'onset_head.0.layer3.1.weight', 'onset_head.1.mlm.bias_hh_l0', 'onset_head.1.mlm.bias_ih_l0_reverse',
             'pitch_head.0.fc1.0.bias', 'pitch_head.0.layer2.1.bias', 'adjoin.0.mlm.bias_hh_l0', 'adjoin.1.output_layer.weight',
             'pitch_head.1.output_layer.bias']


# recurrent matrices of model_complexity 4 (1536 x 384 / 1536 x 176; 5: 2048 x 512 / 2048 x 176): every 64th row of their gradients -- 24 rows each pin the
# hidden-384 streaming forward / backward kernels' saved h and W_hh^T fragment order without 2 MB of fixture
MC4_GSLICES = [('onset_head.1.mlm.weight_hh_l0', 64), ('onset_head.1.mlm.weight_hh_l0_reverse', 64), ('adjoin.0.mlm.weight_ih_l0_reverse', 64),
               ('adjoin.0.mlm.weight_hh_l0', 64)]


# model_complexity 5: the refinement stage's output layer (88 x 1024) as every 4th row
MC5_GKEYS = [k for k in MC4_GKEYS if k != 'adjoin.1.output_layer.weight']
MC5_GSLICES = MC4_GSLICES + [('adjoin.1.output_layer.weight', 4)]


def gen_of_train(name, seed, dim_in, mc, B, T, gkeys=None, gslices=None):
    """Training-mode golden: BatchNorm batch statistics, Dropout disabled (p=0) so the result is
    deterministic; labels given -> the reference's losses and a few gradients."""
    profile = rtools.PianoProfile()
    model = OnsetsFrames(dim_in, profile, 1, mc)
    sd_np = load_weights(model, seed, dim_in=dim_in, in_channels=1, model_complexity=mc, offsets=False)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    feats = features(seed + 100, B, 1, dim_in, T)
    mp = labels(seed + 200, B, T, 0.05)
    on = labels(seed + 300, B, T, 0.01)
    batch = {rtools.KEY_FEATS: torch.from_numpy(feats), rtools.KEY_MULTIPITCH: torch.from_numpy(mp),
             rtools.KEY_ONSETS: torch.from_numpy(on)}
    out = model.run_on_batch(batch)
    loss = out[rtools.KEY_LOSS]
    loss[rtools.KEY_LOSS_TOTAL].backward()
    named = dict(model.named_parameters())
    gkeys = gkeys or ['onset_head.0.layer1.0.weight', 'onset_head.0.layer3.1.weight', 'onset_head.1.mlm.weight_hh_l0',
                      'pitch_head.0.fc1.0.bias', 'adjoin.0.mlm.weight_ih_l0_reverse', 'adjoin.1.output_layer.weight',
                      'pitch_head.1.output_layer.bias']
    rec = dict(seed=seed, dim_in=dim_in, model_complexity=mc, feats=feats, multi_pitch=mp, onsets=on,
               wsum=weight_checksum(sd_np),
               loss_pitch=loss[rtools.KEY_LOSS_PITCH].item(), loss_onsets=loss[rtools.KEY_LOSS_ONSETS].item(),
               loss_total=loss[rtools.KEY_LOSS_TOTAL].item(),
               out_onsets=out[rtools.KEY_ONSETS].numpy(), out_multi_pitch=out[rtools.KEY_MULTIPITCH].numpy())
    for i, k in enumerate(gkeys):
        rec[f'grad_{i}'] = named[k].grad.detach().numpy().copy()
    rec['grad_keys'] = np.array(gkeys)
    if gslices:
        rec['gslice_keys'] = np.array([k for k, _ in gslices])
        rec['gslice_step'] = np.array([st for _, st in gslices])
        for i, (k, st) in enumerate(gslices):
            rec[f'gslice_{i}'] = named[k].grad.detach().numpy()[::st].copy()
    # NOTE: the labels-without-onsets branch (onsetsframes.py:176-178) cannot be recorded: with tensor
    # labels tools.multi_pitch_to_onsets returns an ndarray and LogisticBank.get_loss then fails on
    # `.clone()` (models/common.py:566) -- the reference only works when onsets labels are supplied.
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, 'losses', rec['loss_pitch'], rec['loss_onsets'], rec['loss_total'])


def gen_of2_train(name, seed, dim_in, mc, B, T):
    """Training-mode golden of OnsetsFrames2 as shipped (offset head, detach_heads=True default, model_complexity 3): the
    reference's four losses and a few small gradients (Dropout p = 0)."""
    profile = rtools.PianoProfile()
    model = OnsetsFrames2(dim_in, profile, 1, mc)
    sd_np = load_weights(model, seed, dim_in=dim_in, in_channels=1, model_complexity=mc, offsets=True)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    feats = features(seed + 100, B, 1, dim_in, T)
    mp, on, off = labels(seed + 200, B, T, 0.05), labels(seed + 300, B, T, 0.01), labels(seed + 400, B, T, 0.01)
    batch = {rtools.KEY_FEATS: torch.from_numpy(feats), rtools.KEY_MULTIPITCH: torch.from_numpy(mp),
             rtools.KEY_ONSETS: torch.from_numpy(on), rtools.KEY_OFFSETS: torch.from_numpy(off)}
    out = model.run_on_batch(batch)
    loss = out[rtools.KEY_LOSS]
    loss[rtools.KEY_LOSS_TOTAL].backward()
    named = dict(model.named_parameters())
    gkeys = ['offset_head.0.layer1.0.weight', 'onset_head.0.layer3.1.weight', 'offset_head.1.mlm.bias_hh_l0',
             'onset_head.1.mlm.bias_ih_l0_reverse', 'pitch_head.0.fc1.0.bias', 'pitch_head.0.layer2.1.bias',
             'adjoin.0.mlm.bias_hh_l0', 'adjoin.1.output_layer.bias', 'offset_head.2.output_layer.bias']
    rec = dict(seed=seed, dim_in=dim_in, model_complexity=mc, feats=feats, multi_pitch=mp, onsets=on, offsets=off,
               wsum=weight_checksum(sd_np), loss_keys=np.array(sorted(loss.keys())),
               loss_values=np.array([loss[k].item() for k in sorted(loss.keys())]))
    for i, k in enumerate(gkeys):
        rec[f'grad_{i}'] = named[k].grad.detach().numpy().copy()
    rec['grad_keys'] = np.array(gkeys)
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, dict(zip(rec['loss_keys'], rec['loss_values'])))


def gen_tabcnn(name, seed, dim_in, B, T):
    """BASELINE config 1 (TabCNN + CQT features, CPU): eval output and training loss of the reference's TabCNN with a
    GuitarProfile(num_frets=19) on seed features of a GuitarSet-shape CQT (192 bins)."""
    profile = rtools.GuitarProfile(num_frets=19)
    model = TabCNN(dim_in, profile, 1, 1)
    sd_np = synth_tabcnn_state_dict(seed, dim_in=dim_in, in_channels=1, model_complexity=1, num_groups=6, num_classes=21)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd_np.keys()), (list(ref_sd.keys()), list(sd_np.keys()))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd_np[k].shape), (k, v.shape, sd_np[k].shape)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    model.eval()
    feats = features(seed + 100, B, 1, dim_in, T)
    rng = np.random.default_rng(seed + 200)
    tab = rng.integers(-1, 20, size=(B, 6, T)).astype(np.int64)
    with torch.no_grad():
        pre = model.pre_proc({rtools.KEY_FEATS: torch.from_numpy(feats)})
        raw = model(pre[rtools.KEY_FEATS])[rtools.KEY_TABLATURE]
        out = model.run_on_batch({rtools.KEY_FEATS: torch.from_numpy(feats), rtools.KEY_TABLATURE: torch.from_numpy(tab)})
    rec = dict(seed=seed, dim_in=dim_in, feats=feats, tablature_ref=tab, wsum=weight_checksum(sd_np),
               framed_shape=np.array(pre[rtools.KEY_FEATS].shape), logits=raw.numpy(),
               out_tablature=out[rtools.KEY_TABLATURE].numpy(), loss_total=out[rtools.KEY_LOSS][rtools.KEY_LOSS_TOTAL].item(),
               midi_low=profile.low, midi_high=profile.high, dof_range=profile.get_dof_midi_range())
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, 'logits', rec['logits'].shape, 'loss', rec['loss_total'], 'range', profile.low, profile.high)


def gen_labels_and_cache(name, cache_name, seed, T, n_notes, hop=512, sr=22050):
    """Ground-truth rasterisation (tools/utils.py:1665-1737,2329-2378,2508-2552) of random notes -- some out of the piano's
    range or outside the time grid -- and a feature cache file written by the reference's own save_dict_npz
    (datasets/common.py:259-266)."""
    rng = np.random.default_rng(seed)
    profile = rtools.PianoProfile()
    times = np.arange(T) * hop / float(sr)
    pitches = rng.uniform(15, 115, n_notes)                       # a few below 21 / above 108
    onsets = rng.uniform(-0.5, times[-1] + 0.5, n_notes)
    intervals = np.stack([onsets, onsets + rng.uniform(0.01, 1.5, n_notes)], axis=-1)
    rec = dict(pitches=pitches, intervals=intervals, times=times,
               multi_pitch=rtools.notes_to_multi_pitch(pitches.copy(), intervals.copy(), times, profile),
               multi_pitch_no_off=rtools.notes_to_multi_pitch(pitches.copy(), intervals.copy(), times, profile, include_offsets=False),
               onsets=rtools.notes_to_onsets(pitches.copy(), intervals.copy(), times, profile),
               onsets_amb=rtools.notes_to_onsets(pitches.copy(), intervals.copy(), times, profile, ambiguity=0.05),
               offsets=rtools.notes_to_offsets(pitches.copy(), intervals.copy(), times, profile),
               offsets_amb=rtools.notes_to_offsets(pitches.copy(), intervals.copy(), times, profile, ambiguity=0.05))
    np.savez_compressed(os.path.join(OUT, name), **rec)
    feats = features(seed + 1, 1, 1, 40, 12)[0]
    rtools.save_dict_npz(os.path.join(OUT, cache_name), {rtools.KEY_FS: sr, rtools.KEY_HOP: hop, rtools.KEY_FEATS: feats})
    rtools.save_dict_npz(os.path.join(OUT, cache_name.replace('.npz', '_none.npz')), {rtools.KEY_FS: sr, rtools.KEY_HOP: hop, rtools.KEY_FEATS: None})
    print(name, {k: v.shape for k, v in rec.items()}, 'active cells', int(rec['multi_pitch'].sum()))


def gen_labels_edges(name, hop=512, sr=22050, T=50):
    """Rasterisation edge cases: onsets / offsets exactly on grid points, on the extension point one hop past the last frame,
    before the first frame, beyond the end, zero-length notes, notes on the pitch-range borders and half-way pitches."""
    profile = rtools.PianoProfile()
    times = np.arange(T) * hop / float(sr)
    h = hop / float(sr)
    end = times[-1] + h                      # the extension point
    iv = [[times[3], times[7]], [times[3], times[3]], [0.0, 0.0], [-1.0, 0.0], [-1.0, -0.5], [-0.2, times[2] + 0.3 * h],
          [times[-1], end], [end, end], [end, end + 1.0], [times[-2], end + 1.0], [end + 0.1, end + 0.2], [times[10] + 0.5 * h, times[10] + 0.6 * h],
          [times[20], times[20] + h], [times[20] - 1e-12, times[25] + 1e-12], [times[30], times[29]], [0.0, end], [times[40], times[44]],
          [times[41], times[42]]]
    pitches = np.array([60, 61, 21, 108, 50, 20.5, 108.49, 70, 71, 72, 73, 20.4, 107.5, 64.5, 65, 30, 90, 90], dtype=np.float64)[:len(iv)]
    intervals = np.array(iv, dtype=np.float64)
    rec = dict(pitches=pitches, intervals=intervals, times=times)
    for key, fn, kw in (('multi_pitch', rtools.notes_to_multi_pitch, {}), ('multi_pitch_no_off', rtools.notes_to_multi_pitch, {'include_offsets': False}),
                        ('onsets', rtools.notes_to_onsets, {}), ('onsets_amb', rtools.notes_to_onsets, {'ambiguity': 0.05}),
                        ('offsets', rtools.notes_to_offsets, {}), ('offsets_amb', rtools.notes_to_offsets, {'ambiguity': 0.05})):
        rec[key] = fn(pitches.copy(), intervals.copy(), times, profile, **kw)
    fp, fi = rtools.filter_notes(pitches.copy(), intervals.copy(), profile, min_time=0.0, max_time=end)
    rec['filtered_pitches'], rec['filtered_intervals'] = fp, fi
    irregular = np.concatenate([times[:10], times[10:20] + 3.0, [times[19] + 3.0 + 0.7 * h]])
    rec['irregular_times'] = irregular
    rec['irregular_hop'] = np.array(rtools.estimate_hop_length(irregular))
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, {k: v.shape for k, v in rec.items()}, 'active cells', int(rec['multi_pitch'].sum()))


def gen_feature_bookkeeping(name):
    """Frame / sample bookkeeping of the reference's STFT and MelSpec FeatureModules (features/common.py:41-150,232-321,
    stft.py, mel.py; waveform.py for the non-centred framing) -- pure arithmetic, runs under the librosa stub -- and their
    dB post-processing on a fixed array with the documented librosa.amplitude_to_db / power_to_db formulas plugged into the stub."""
    from amt_tools.features import STFT, MelSpec

    def _to_db(S, ref, amin, top_db, mult):
        S = np.asarray(S)
        ref_v = ref(S) if callable(ref) else ref
        log = mult * np.log10(np.maximum(amin, S)) - mult * np.log10(np.maximum(amin, ref_v))
        return np.maximum(log, log.max() - top_db)

    sys.modules['librosa.core'].amplitude_to_db = lambda S, ref=1.0, amin=1e-5, top_db=80.0: _to_db(np.abs(S) ** 2, (lambda P: ref(np.sqrt(P)) ** 2) if callable(ref) else ref ** 2, amin ** 2, top_db, 10.0)
    sys.modules['librosa.core'].power_to_db = lambda S, ref=1.0, amin=1e-10, top_db=80.0: _to_db(S, ref, amin, top_db, 10.0)
    sys.modules['librosa'].core = sys.modules['librosa.core']
    sys.modules['librosa'].frames_to_time = lambda frames, sr=22050, hop_length=512, n_fft=None: (np.asarray(frames) * hop_length + (n_fft // 2 if n_fft else 0)) / float(sr)
    lengths = np.array([0, 1, 2, 511, 512, 513, 1023, 1024, 2047, 2048, 2049, 4095, 4096, 22050, 319999, 320000])
    frames = np.array([0, 1, 2, 3, 10, 625])
    rec = dict(lengths=lengths, frames=frames)
    mods = {'stft_c': STFT(sample_rate=22050, hop_length=512, n_fft=2048), 'stft_nc': STFT(sample_rate=16000, hop_length=256, n_fft=1024, win_length=800, center=False),
            'mel_c': MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048), 'mel_nc': MelSpec(sample_rate=16000, hop_length=512, n_mels=64, n_fft=2048, center=False)}
    for key, m in mods.items():
        rec[key + '_expected_frames'] = np.array([m.get_expected_frames(np.zeros(int(n))) for n in lengths])
        sr_ = [m.get_sample_range(int(f)) for f in frames]
        rec[key + '_sample_range_min'] = np.array([int(np.min(r)) for r in sr_])
        rec[key + '_sample_range_max'] = np.array([int(np.max(r)) for r in sr_])
        rec[key + '_sample_range_len'] = np.array([len(r) for r in sr_])
        rec[key + '_num_samples_required'] = np.array(m.get_num_samples_required())
        rec[key + '_feature_size'] = np.array(m.get_feature_size())
        rec[key + '_num_channels'] = np.array(m.get_num_channels())
        rec[key + '_times'] = np.asarray(m.get_times(np.zeros(5000)), dtype=np.float64)
        rec[key + '_name'] = np.array(m.features_name())
    rng = np.random.default_rng(3)
    S = np.abs(rng.standard_normal((5, 12))) ** 3 * 10.0
    rec['db_in'] = S
    rec['stft_post'] = mods['stft_c'].post_proc(S.copy())                  # magnitude -> dB -> scaled, channel axis
    rec['mel_post'] = mods['mel_c'].post_proc(S.copy())                    # power -> dB -> scaled
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, {k: (v.tolist() if v.size < 20 else v.shape) for k, v in rec.items() if 'mel_c' in k})


def gen_notes(name, seed, T, p_on, p_mp, with_onsets=True, hop=512, sr=22050, times_dtype=np.float64, inhibition_window=None,
              minimum_duration=None):
    rng = np.random.default_rng(seed)
    profile = rtools.PianoProfile()
    est = NoteTranscriber(profile=profile, inhibition_window=inhibition_window, minimum_duration=minimum_duration)
    # sticky activations so notes have duration
    mp = np.zeros((88, T), dtype=np.float32)
    state = rng.random(88) < p_mp
    for t in range(T):
        flip = rng.random(88)
        state = np.where(state, flip > 0.15, flip < p_mp * 0.3)
        mp[:, t] = state
    on = (rng.random((88, T)) < p_on).astype(np.float32)
    times = (np.arange(T) * hop / float(sr)).astype(times_dtype)     # run_offline casts the grid to float32 (inference.py:36)
    raw = {rtools.KEY_MULTIPITCH: mp.copy(), rtools.KEY_TIMES: times.copy()}
    if with_onsets:
        raw[rtools.KEY_ONSETS] = on.copy()
    notes = est.estimate(raw)
    rec = dict(multi_pitch=mp, times=times, notes=np.asarray(notes, dtype=np.float64), with_onsets=int(with_onsets),
               inhibition_window=np.array(-1.0 if inhibition_window is None else inhibition_window),
               minimum_duration=np.array(-1.0 if minimum_duration is None else minimum_duration))
    if with_onsets:
        rec['onsets'] = on
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, 'notes', rec['notes'].shape)


def gen_rms_norm(name):
    """amt_tools.tools.rms_norm (tools/utils.py:2789-2814) on float32 and float64 clips of several scales, a silent clip and a one-sample
    clip: inputs (a seed per clip, not the samples) + the reference's outputs, downsampled to every 97th sample plus the exact RMS."""
    rec = {'seeds': np.arange(5), 'lengths': np.array([100001, 40000, 7, 1, 5000]), 'scales': np.array([0.01, 1.0, 30.0, 2.5, 0.0]), 'stride': 97}
    for i, (n, sc) in enumerate(zip(rec['lengths'], rec['scales'])):
        x64 = np.random.default_rng(700 + i).standard_normal(int(n)) * sc
        for dt in (np.float32, np.float64):
            x = x64.astype(dt)
            y = rtools.rms_norm(x)
            tag = f'{i}_{np.dtype(dt).name}'
            rec[f'out_{tag}'] = np.asarray(y)[::97].copy()
            rec[f'dtype_{tag}'] = str(np.asarray(y).dtype)
            rec[f'sumsq_{tag}'] = np.float64(np.sum(np.asarray(y, dtype=np.float64) ** 2))
    np.savez_compressed(os.path.join(OUT, name), **rec)
    print(name, 'rms_norm', len(rec))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'rms_norm':
        gen_rms_norm('rms_norm.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'labels_edges':
        gen_labels_edges('labels_edges.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'notes_opts':
        gen_notes('notes_inhibit.npz', 36, 300, 0.0, 0.08, False, inhibition_window=0.08)
        gen_notes('notes_inhibit_onsets.npz', 37, 300, 0.02, 0.08, True, inhibition_window=0.08)      # no effect with onsets given (transcribe.py:463-468)
        gen_notes('notes_mindur.npz', 38, 300, 0.02, 0.08, True, minimum_duration=0.06)
        gen_notes('notes_mindur0.npz', 39, 300, 0.0, 0.08, False, minimum_duration=0.0)
        gen_notes('notes_inhibit_mindur.npz', 40, 400, 0.0, 0.1, False, inhibition_window=0.05, minimum_duration=0.05)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'notes_f32':
        gen_notes('notes_f32times.npz', 35, 625, 0.01, 0.05, True, sr=16000, times_dtype=np.float32)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'mc4':
        gen_of_eval('of1_mc4_eval.npz', OnsetsFrames, seed=15, dim_in=229, in_channels=1, mc=4, B=2, T=24, offsets=False)
        gen_of_eval('of2_mc4_hcqt_eval.npz', OnsetsFrames2, seed=16, dim_in=72, in_channels=3, mc=4, B=1, T=20, offsets=True)
        gen_of_train('of1_mc4_train.npz', seed=23, dim_in=229, mc=4, B=2, T=16, gkeys=MC4_GKEYS, gslices=MC4_GSLICES)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'mc5':
        gen_of_eval('of1_mc5_eval.npz', OnsetsFrames, seed=17, dim_in=229, in_channels=1, mc=5, B=2, T=24, offsets=False)
        gen_of_eval('of2_mc5_hcqt_eval.npz', OnsetsFrames2, seed=18, dim_in=72, in_channels=3, mc=5, B=1, T=20, offsets=True)
        gen_of_train('of1_mc5_train.npz', seed=24, dim_in=229, mc=5, B=2, T=16, gkeys=MC5_GKEYS, gslices=MC5_GSLICES)
        sys.exit(0)
    gen_of_eval('of1_eval.npz', OnsetsFrames, seed=11, dim_in=229, in_channels=1, mc=2, B=2, T=40, offsets=False)
    gen_of_eval('of1_mc4_eval.npz', OnsetsFrames, seed=15, dim_in=229, in_channels=1, mc=4, B=2, T=24, offsets=False)
    gen_of_eval('of2_mc4_hcqt_eval.npz', OnsetsFrames2, seed=16, dim_in=72, in_channels=3, mc=4, B=1, T=20, offsets=True)
    gen_of_train('of1_mc4_train.npz', seed=23, dim_in=229, mc=4, B=2, T=16, gkeys=MC4_GKEYS, gslices=MC4_GSLICES)
    gen_of_eval('of1_mc5_eval.npz', OnsetsFrames, seed=17, dim_in=229, in_channels=1, mc=5, B=2, T=24, offsets=False)
    gen_of_eval('of2_mc5_hcqt_eval.npz', OnsetsFrames2, seed=18, dim_in=72, in_channels=3, mc=5, B=1, T=20, offsets=True)
    gen_of_train('of1_mc5_train.npz', seed=24, dim_in=229, mc=5, B=2, T=16, gkeys=MC5_GKEYS, gslices=MC5_GSLICES)
    gen_of_eval('of1_hcqt_eval.npz', OnsetsFrames, seed=12, dim_in=72, in_channels=6, mc=2, B=1, T=33, offsets=False)
    gen_of_eval('of2_eval.npz', OnsetsFrames2, seed=13, dim_in=229, in_channels=1, mc=3, B=1, T=24, offsets=True)
    gen_of_eval('of2_mc2_eval.npz', OnsetsFrames2, seed=14, dim_in=229, in_channels=1, mc=2, B=2, T=36, offsets=True)
    gen_of_train('of1_train.npz', seed=21, dim_in=229, mc=2, B=2, T=24)
    gen_of2_train('of2_train.npz', seed=22, dim_in=229, mc=3, B=2, T=24)
    gen_tabcnn('tabcnn_eval.npz', seed=41, dim_in=192, B=2, T=30)
    gen_labels_and_cache('labels.npz', 'feature_cache_ref.npz', seed=51, T=200, n_notes=60)
    gen_labels_edges('labels_edges.npz')
    gen_rms_norm('rms_norm.npz')
    gen_feature_bookkeeping('feature_bookkeeping.npz')
    gen_notes('notes_dense.npz', 31, 300, 0.02, 0.08, True)
    gen_notes('notes_sparse.npz', 32, 625, 0.002, 0.01, True)
    gen_notes('notes_noonsets.npz', 33, 200, 0.0, 0.06, False)
    gen_notes('notes_empty.npz', 34, 64, 0.0, 0.0, True)
    gen_notes('notes_f32times.npz', 35, 625, 0.01, 0.05, True, sr=16000, times_dtype=np.float32)
    gen_notes('notes_inhibit.npz', 36, 300, 0.0, 0.08, False, inhibition_window=0.08)
    gen_notes('notes_inhibit_onsets.npz', 37, 300, 0.02, 0.08, True, inhibition_window=0.08)
    gen_notes('notes_mindur.npz', 38, 300, 0.02, 0.08, True, minimum_duration=0.06)
    gen_notes('notes_mindur0.npz', 39, 300, 0.0, 0.08, False, minimum_duration=0.0)
    gen_notes('notes_inhibit_mindur.npz', 40, 400, 0.0, 0.1, False, inhibition_window=0.05, minimum_duration=0.05)
