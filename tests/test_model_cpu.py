"""Host logic of the TranscriptionModel mirror on CPU (torch path = training / autograd path): the
reference's golden vectors (eval outputs, training losses and gradients), dict contract, pickling."""
import io

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_grad_slices
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict, of_state_dict_shapes


def _model(g, **kw):
    model = OnsetsFrames(int(g['dim_in']), tools.PianoProfile(), int(g.get('in_channels', 1)), int(g['model_complexity']), **kw)
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=int(g.get('in_channels', 1)),
                          model_complexity=int(g['model_complexity']))
    assert list(model.state_dict().keys()) == list(sd.keys())          # same keys, same order as the reference
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(s) for k, s in of_state_dict_shapes(
        dim_in=int(g['dim_in']), in_channels=int(g.get('in_channels', 1)), model_complexity=int(g['model_complexity'])).items()}
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return model


def test_eval_run_on_batch_matches_reference_on_cpu():
    g = load_golden('of1_eval.npz')
    model = _model(g)
    model.eval()
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])}
    with torch.no_grad():
        out = model.run_on_batch(batch)
    assert set(out.keys()) == {tools.KEY_ONSETS, tools.KEY_MULTIPITCH, tools.KEY_TIMES}
    for key in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < 2e-5
        assert np.all((out[key].numpy() == g['out_' + key]) | near)
    assert batch[tools.KEY_FEATS].shape == (2, 1, 229, 40)            # caller's batch untouched


@pytest.mark.parametrize('name', ['of2_eval.npz', 'of2_mc2_eval.npz', 'of2_mc4_hcqt_eval.npz', 'of2_mc5_hcqt_eval.npz'])
def test_onsetsframes2_matches_reference_on_cpu(name):
    """OnsetsFrames2 (offset head, detach_heads, model_complexity 3 and 2): same state_dict keys as the reference, same logits,
    same outputs (offsets as probabilities, onsetsframes.py:323-325)."""
    g = load_golden(name)
    mc = int(g['model_complexity'])
    ic = int(g['in_channels'])
    model = OnsetsFrames2(int(g['dim_in']), tools.PianoProfile(), ic, mc)
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=ic, model_complexity=mc, offsets=True)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.eval()
    assert model.model_name() == 'OnsetsFrames2' and model.detach_heads is True
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])}
    with torch.no_grad():
        raw = model(model.pre_proc(dict(batch))[tools.KEY_FEATS])
        out = model.run_on_batch(batch)
    for key in (tools.KEY_ONSETS, tools.KEY_OFFSETS, tools.KEY_MULTIPITCH):
        assert np.abs(raw[key].numpy() - g['logits_' + key]).max() < 2e-5, key
    assert set(out.keys()) == {tools.KEY_ONSETS, tools.KEY_OFFSETS, tools.KEY_MULTIPITCH, tools.KEY_TIMES}
    assert np.abs(out[tools.KEY_OFFSETS].numpy() - g['out_offsets']).max() < 1e-5
    for key in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < 2e-5
        assert np.all((out[key].numpy() == g['out_' + key]) | near)
    # training: the offsets term is part of the total loss; offsets labels derived from the multi-pitch labels when absent
    model.train()
    T = g['feats'].shape[-1]
    mp = (np.random.default_rng(0).random((g['feats'].shape[0], 88, T)) < 0.05).astype(np.float32)
    res = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_MULTIPITCH: torch.from_numpy(mp),
                              tools.KEY_ONSETS: torch.from_numpy(tools.multi_pitch_to_onsets(mp.copy()))})
    loss = res[tools.KEY_LOSS]
    assert set(loss.keys()) == {tools.KEY_LOSS_PITCH, tools.KEY_LOSS_ONSETS, tools.KEY_LOSS_OFFSETS, tools.KEY_LOSS_TOTAL}
    total = loss[tools.KEY_LOSS_PITCH] + loss[tools.KEY_LOSS_ONSETS] + loss[tools.KEY_LOSS_OFFSETS]
    assert abs(loss[tools.KEY_LOSS_TOTAL].item() - total.item()) < 1e-4
    off = tools.multi_pitch_to_offsets(mp.copy())
    assert off.shape == mp.shape and off[..., -1].sum() == mp[..., -1].sum() and set(np.unique(off)) <= {0.0, 1.0}


@pytest.mark.parametrize('name', ['of1_train.npz', 'of1_mc4_train.npz', 'of1_mc5_train.npz'])
def test_training_step_matches_reference_losses_and_grads(name):
    g = load_golden(name)
    model = _model(g)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_MULTIPITCH: torch.from_numpy(g['multi_pitch']),
             tools.KEY_ONSETS: torch.from_numpy(g['onsets'])}
    out = model.run_on_batch(batch)
    loss = out[tools.KEY_LOSS]
    assert set(loss.keys()) == {tools.KEY_LOSS_PITCH, tools.KEY_LOSS_ONSETS, tools.KEY_LOSS_TOTAL}
    assert abs(loss[tools.KEY_LOSS_PITCH].item() - float(g['loss_pitch'])) < 1e-3
    assert abs(loss[tools.KEY_LOSS_ONSETS].item() - float(g['loss_onsets'])) < 1e-3
    loss[tools.KEY_LOSS_TOTAL].backward()
    named = dict(model.named_parameters())
    for i, k in enumerate(g['grad_keys']):
        ref = g[f'grad_{i}']
        assert np.abs(named[str(k)].grad.numpy() - ref).max() / max(1e-6, np.abs(ref).max()) < 2e-3, k
    for k, st, ref in golden_grad_slices(g):
        assert np.abs(named[k].grad.numpy()[::st] - ref).max() / max(1e-6, np.abs(ref).max()) < 2e-3, k
    # onsets derived from the multi-pitch labels when none are given (intended behaviour of onsetsframes.py:176-178)
    out2 = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_MULTIPITCH: torch.from_numpy(g['multi_pitch'])})
    assert torch.isfinite(out2[tools.KEY_LOSS][tools.KEY_LOSS_ONSETS])


def test_pickle_roundtrip_and_attributes():
    g = load_golden('of1_eval.npz')
    model = _model(g)
    for attr in ('dim_in', 'profile', 'in_channels', 'model_complexity', 'frame_width', 'device', 'iter', 'frontend'):
        assert hasattr(model, attr)
    assert model.model_name() == 'OnsetsFrames' and model.frame_width == 1 and len(model.frontend) == 0
    model.iter += 3
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    clone = torch.load(buf, weights_only=False)
    assert clone.iter == 3
    for (k, a), (_, b) in zip(model.state_dict().items(), clone.state_dict().items()):
        assert torch.equal(a, b), k
    clone.change_device('cpu')
    assert clone.device == 'cpu'


def test_dict_plumbing_semantics():
    t = {'a': np.ones((2, 3), dtype=np.float64), 'n': {'b': np.zeros(4)}, 's': 'x'}
    f = tools.dict_to_dtype(t, 'float32')
    assert f['a'].dtype == np.float32 and t['a'].dtype == np.float64 and f['s'] == 'x'
    tt = tools.dict_unsqueeze(tools.dict_to_tensor(f))
    assert tt['a'].shape == (1, 2, 3) and tt['n']['b'].shape == (1, 4)
    back = tools.dict_squeeze(tools.dict_to_array(tt), dim=0)
    assert back['a'].shape == (2, 3) and isinstance(back['a'], np.ndarray)
    x = torch.tensor([[0.2, 0.5, 0.7]])
    assert torch.equal(tools.threshold_activations(x.clone(), 0.5), torch.tensor([[0., 1., 1.]]))
    mp = np.array([[0, 1, 1, 0, 1]], dtype=np.float32)
    np.testing.assert_array_equal(tools.multi_pitch_to_onsets(mp), [[0, 1, 0, 0, 1]])
    np.testing.assert_array_equal(tools.multi_pitch_to_onsets(torch.from_numpy(mp)).numpy(), [[0, 1, 0, 0, 1]])


def test_run_offline_contract_on_cpu():
    from amt_tools_amd.inference import run_offline, run_offline_batched
    from amt_tools_amd.transcribe import NoteTranscriber
    g = load_golden('of1_eval.npz')
    model = _model(g)
    model.eval()
    track = {tools.KEY_TRACK: 'x', tools.KEY_FEATS: g['feats'][0].astype(np.float64), tools.KEY_TIMES: g['out_times'][0]}
    pred = run_offline(track, model, NoteTranscriber(tools.PianoProfile()))
    assert pred[tools.KEY_ONSETS].shape == (88, 40) and isinstance(pred[tools.KEY_ONSETS], np.ndarray)
    assert pred[tools.KEY_NOTES].ndim == 2 and pred[tools.KEY_NOTES].shape[1] == 3
    near = np.abs(g['logits_multi_pitch'][0].T) < 2e-5
    assert np.all((pred[tools.KEY_MULTIPITCH] == g['out_multi_pitch'][0]) | near)
    # batched driver: every clip's result equals the single-clip result; shards partition the clips
    res0 = run_offline_batched(g['feats'], model, batch_size=1, rank=0, world=2)
    res1 = run_offline_batched(g['feats'], model, batch_size=1, rank=1, world=2)
    assert sorted(res0) == [0] and sorted(res1) == [1]
    np.testing.assert_array_equal(res0[0][tools.KEY_ONSETS], pred[tools.KEY_ONSETS])


def test_config1_tabcnn_matches_reference_on_cpu():
    """BASELINE config 1 (TabCNN on CQT-shaped features, CPU plumbing): the mirror reproduces the REAL reference's framing,
    logits, tablature and loss (tests/golden/tabcnn_eval.npz) and the GuitarProfile geometry."""
    from amt_tools_amd.models import TabCNN
    from amt_tools_amd.synth import synth_tabcnn_state_dict
    g = load_golden('tabcnn_eval.npz')
    profile = tools.GuitarProfile(num_frets=19)
    assert (profile.low, profile.high, profile.get_num_dofs(), profile.num_pitches) == (int(g['midi_low']), int(g['midi_high']), 6, 20)
    np.testing.assert_array_equal(profile.get_dof_midi_range(), g['dof_range'])
    model = TabCNN(int(g['dim_in']), profile, 1, 1)
    sd = synth_tabcnn_state_dict(int(g['seed']), dim_in=int(g['dim_in']))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.eval()
    assert model.frame_width == 9 and model.model_name() == 'TabCNN'
    feats = torch.from_numpy(g['feats'])
    with torch.no_grad():
        pre = model.pre_proc({tools.KEY_FEATS: feats})
        assert tuple(pre[tools.KEY_FEATS].shape) == tuple(g['framed_shape'])          # (B, T, C, F, 9)
        raw = model(pre[tools.KEY_FEATS])[tools.KEY_TABLATURE]
        out = model.run_on_batch({tools.KEY_FEATS: feats, tools.KEY_TABLATURE: torch.from_numpy(g['tablature_ref'])})
    assert np.abs(raw.numpy() - g['logits']).max() < 2e-5
    np.testing.assert_array_equal(out[tools.KEY_TABLATURE].numpy(), g['out_tablature'])
    assert abs(out[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].item() - float(g['loss_total'])) < 1e-4
    # framing helper: centre zero padding, one window per frame
    x = np.arange(5, dtype=np.float32)[None]
    fr = tools.framify_activations(x, 3)
    np.testing.assert_array_equal(fr[0], [[0, 0, 1], [0, 1, 2], [1, 2, 3], [2, 3, 4], [3, 4, 0]])
    assert tools.framify_activations(x, 9, pad=False).shape == (1, 1, 9)
    assert tools.note_to_midi(['E2', 'A2', 'D3', 'G3', 'B3', 'E4']) == [40, 45, 50, 55, 59, 64] and tools.note_to_midi('Bb3') == 58


def test_onsetsframes2_training_step_matches_reference_losses_and_grads():
    """OnsetsFrames2 as shipped (model_complexity 3, offset head, detach_heads) in training mode on the CPU: the four losses and
    a few gradients recorded from the REAL reference classes (tests/golden/of2_train.npz, tools/gen_golden.py)."""
    from amt_tools_amd.models import OnsetsFrames2
    from amt_tools_amd.synth import synth_state_dict
    g = load_golden('of2_train.npz')
    mc = int(g['model_complexity'])
    model = OnsetsFrames2(int(g['dim_in']), tools.PianoProfile(), 1, mc)
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=1, model_complexity=mc, offsets=True)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_MULTIPITCH: torch.from_numpy(g['multi_pitch']),
             tools.KEY_ONSETS: torch.from_numpy(g['onsets']), tools.KEY_OFFSETS: torch.from_numpy(g['offsets'])}
    loss = model.run_on_batch(batch)[tools.KEY_LOSS]
    assert sorted(loss.keys()) == [str(k) for k in g['loss_keys']]
    for k, v in zip(g['loss_keys'], g['loss_values']):
        assert abs(loss[str(k)].item() - float(v)) < 1e-3 * max(1.0, abs(float(v))), k
    loss[tools.KEY_LOSS_TOTAL].backward()
    named = dict(model.named_parameters())
    for i, k in enumerate(g['grad_keys']):
        ref = g[f'grad_{i}']
        if str(k).endswith('.0.bias') and '.layer' in str(k):
            continue
        assert np.abs(named[str(k)].grad.numpy() - ref).max() / max(1e-6, np.abs(ref).max()) < 2e-3, k


def test_frame_times_come_from_the_front_end_module_and_safe_globals_cover_a_pickled_model(tmp_path):
    """run_offline_batched(decode_notes=True) without `times` derives the grid from the model's front-end (hop / sample rate of
    THAT module, not a hard-coded 512 / 22050) and refuses to guess without one; register_safe_globals() lets torch >= 2.6 load a
    whole pickled model the way amt_tools/train.py:106 does on resume."""
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.inference import _frame_times
    from amt_tools_amd.models import OnsetsFrames
    model = OnsetsFrames(40, tools.PianoProfile(), 1, 2)
    with pytest.raises(ValueError):
        _frame_times(model, 5)
    model.frontend = torch.nn.Sequential(MelSpec(sample_rate=16000, hop_length=256, n_mels=40).frontend())
    np.testing.assert_array_equal(_frame_times(model, 4), np.arange(4) * 256 / 16000.0)
    path = str(tmp_path / 'model.pt')
    torch.save(model, path)
    tools.register_safe_globals()
    loaded = torch.load(path)                       # weights_only default of the installed torch
    assert isinstance(loaded, OnsetsFrames) and loaded.frontend[0].module.hop_length == 256


def test_strict_training_switch_turns_a_fallback_into_an_error(monkeypatch):
    from amt_tools_amd import autograd as ag
    monkeypatch.setenv('AMTX_STRICT_TRAINING', '1')
    with pytest.raises(RuntimeError, match='AMTX_STRICT_TRAINING'):
        ag.note_fallback('somewhere', 'some shape')
    monkeypatch.delenv('AMTX_STRICT_TRAINING')
    ag.reset_fallbacks()
    ag.note_fallback('somewhere', 'some shape')
    assert ag.fallbacks() == {'somewhere': ('some shape', 1)}
    ag.reset_fallbacks()


def test_default_precision_of_the_drop_in_classes_is_the_compliant_one():
    """Round 6: a user who only swaps the imports gets the engine mode that is inside north_star's 1e-4 (`x3`); the bf16
    throughput mode of BASELINE config 2 is an explicit opt-in.  The keyword does not exist in the reference and changes nothing on a CPU."""
    from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
    assert OnsetsFrames(40, tools.PianoProfile(), 1, 2).precision == 'x3'
    assert OnsetsFrames2(40, tools.PianoProfile(), 1, 2).precision == 'x3'
    assert OnsetsFrames(40, tools.PianoProfile(), 1, 2, precision='bf16').precision == 'bf16'
