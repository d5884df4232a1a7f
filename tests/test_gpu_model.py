"""GPU parity of the whole Onsets & Frames engine against the golden vectors recorded from the REAL
reference classes (tests/golden/*.npz) and against the oracle on fresh inputs.

Tolerances (stated per north_star): precision 'x3' -> raw logits within 1e-4 of the CPU reference;
precision 'bf16' -> raw logits within 0.06 (bf16 operand rounding through six dense layers and two
recurrences).  The thresholded piano rolls must agree with the reference everywhere except cells whose
reference logit lies inside the tolerance band around the decision boundary."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from conftest import load_golden                               # noqa: E402
from amt_tools_amd import tools                                # noqa: E402
from amt_tools_amd.synth import synth_state_dict, synth_clip   # noqa: E402

TOL = {'x3': 1e-4, 'bf16': 6e-2}


def _model(g, precision):
    from amt_tools_amd.models import OnsetsFrames
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=int(g['in_channels']),
                          model_complexity=int(g['model_complexity']))
    model = OnsetsFrames(int(g['dim_in']), tools.PianoProfile(), int(g['in_channels']), int(g['model_complexity']),
                         device='cuda:0', precision=precision)
    missing = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert not missing.missing_keys and not missing.unexpected_keys        # reference state_dict keys load as they are
    model.change_device()
    model.eval()
    return model


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
@pytest.mark.parametrize('name', ['of1_eval.npz', 'of1_hcqt_eval.npz'])
def test_engine_matches_reference_golden(name, precision):
    g = load_golden(name)
    model = _model(g, precision)
    tol = TOL[precision]
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])}
    with torch.no_grad():
        out = model.run_on_batch(batch)
        logits = model.engine_logits(torch.from_numpy(g['feats']).cuda())
    for key in ('onsets', 'multi_pitch', 'pitch_head'):
        err = np.abs(logits[key].cpu().numpy() - g['logits_' + key]).max()
        assert err < tol, (key, err)
    for key in ('onsets', 'multi_pitch'):
        got = out[key].cpu().numpy()
        assert got.shape == g['out_' + key].shape and got.dtype == np.float32
        assert set(np.unique(got)) <= {0.0, 1.0}
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < tol
        assert np.all((got == g['out_' + key]) | near)
        if precision == 'x3':
            assert (got != g['out_' + key]).mean() < 1e-3
    np.testing.assert_array_equal(out[tools.KEY_TIMES].cpu().numpy(), g['out_times'])
    assert tools.KEY_FEATS in batch and batch[tools.KEY_FEATS].device.type == 'cpu'    # caller's batch untouched


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
def test_onsetsframes2_engine_matches_reference_golden(precision):
    """OnsetsFrames2 (offset head) at model_complexity 2 through the HIP engine vs the real reference's vectors."""
    from amt_tools_amd.models import OnsetsFrames2
    g = load_golden('of2_mc2_eval.npz')
    sd = synth_state_dict(int(g['seed']), dim_in=229, in_channels=1, model_complexity=2, offsets=True)
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, 2, device='cuda:0', precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    tol = TOL[precision]
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])})
        logits = model.engine_logits(torch.from_numpy(g['feats']).cuda())
    for key in ('onsets', 'offsets', 'multi_pitch', 'pitch_head'):
        err = np.abs(logits[key].cpu().numpy() - g['logits_' + key]).max()
        assert err < tol, (key, err)
    assert set(out.keys()) == {tools.KEY_ONSETS, tools.KEY_OFFSETS, tools.KEY_MULTIPITCH, tools.KEY_TIMES}
    assert np.abs(out[tools.KEY_OFFSETS].cpu().numpy() - g['out_offsets']).max() < (1e-4 if precision == 'x3' else 2e-2)
    for key in ('onsets', 'multi_pitch'):
        got = out[key].cpu().numpy()
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < tol
        assert np.all((got == g['out_' + key]) | near)


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
def test_onsetsframes2_default_complexity_3_engine_matches_reference_golden(precision):
    """OnsetsFrames2 as the reference ships it (model_complexity 3: 48/48/96-channel convolutions, LSTM hidden 256, offset head,
    onsetsframes.py:199-233) through the HIP engine vs vectors recorded from the real reference classes."""
    from amt_tools_amd.models import OnsetsFrames2
    g = load_golden('of2_eval.npz')
    assert int(g['model_complexity']) == 3
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=1, model_complexity=3, offsets=True)
    model = OnsetsFrames2(int(g['dim_in']), tools.PianoProfile(), 1, device='cuda:0', precision=precision)
    assert model.model_complexity == 3
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    tol = TOL[precision]
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])})
        logits = model.engine_logits(torch.from_numpy(g['feats']).cuda())
    for key in ('onsets', 'offsets', 'multi_pitch', 'pitch_head'):
        err = np.abs(logits[key].cpu().numpy() - g['logits_' + key]).max()
        assert err < tol, (key, err)
    assert np.abs(out[tools.KEY_OFFSETS].cpu().numpy() - g['out_offsets']).max() < (1e-4 if precision == 'x3' else 2e-2)
    for key in ('onsets', 'multi_pitch'):
        got = out[key].cpu().numpy()
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < tol
        assert np.all((got == g['out_' + key]) | near)


@pytest.mark.parametrize('precision', ['x3', 'bf16', 'f16'])
@pytest.mark.parametrize('name', ['of1_mc4_eval.npz', 'of2_mc4_hcqt_eval.npz', 'of1_mc5_eval.npz', 'of2_mc5_hcqt_eval.npz'])
def test_complexity_4_engine_matches_reference_golden(name, precision):
    """model_complexity 4 (onsetsframes.py:358-364, 40-41: 64/64/128-channel convolutions, fc 1024, LSTM hidden 384) and 5 (80/80/160, fc
    1280, hidden 512) through the HIP engine against vectors recorded from the real reference classes: OnsetsFrames on one-channel mel
    features and OnsetsFrames2 (offset head) on a three-channel HCQT shape."""
    import amt_tools_amd.models as M
    g = load_golden(name)
    mc = int(g['model_complexity'])
    assert mc == (5 if 'mc5' in name else 4)
    offsets = name.startswith('of2')
    cls = M.OnsetsFrames2 if offsets else M.OnsetsFrames
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=int(g['in_channels']), model_complexity=mc, offsets=offsets)
    model = cls(int(g['dim_in']), tools.PianoProfile(), int(g['in_channels']), mc, device='cuda:0', precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    tol = {'x3': 1e-4, 'bf16': 6e-2, 'f16': 1e-2}[precision]
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_TIMES: torch.from_numpy(g['out_times'])})
        logits = model.engine_logits(torch.from_numpy(g['feats']).cuda())
    for key in ('onsets', 'multi_pitch', 'pitch_head') + (('offsets',) if offsets else ()):
        err = np.abs(logits[key].cpu().numpy() - g['logits_' + key]).max()
        assert err < tol, (key, err)
    if offsets:
        assert np.abs(out[tools.KEY_OFFSETS].cpu().numpy() - g['out_offsets']).max() < (1e-4 if precision == 'x3' else 2e-2)
    for key in ('onsets', 'multi_pitch'):
        near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < tol
        assert np.all((out[key].cpu().numpy() == g['out_' + key]) | near)


@pytest.mark.parametrize('mc', [4, 5])
def test_complexity_4_engine_vs_oracle_on_ragged_batches(mc):
    """model_complexity 4 and 5 against the oracle on fresh inputs whose batch and frame counts do not fill the kernels' tiles (x3)."""
    from oracle import model_ref
    from amt_tools_amd.models import OnsetsFrames
    sd = synth_state_dict(12, dim_in=229, in_channels=1, model_complexity=mc)
    model = OnsetsFrames(229, tools.PianoProfile(), 1, mc, device='cuda:0', precision='x3')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    sdt = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(7)
    for B, T in ((1, 1), (3, 17), (17, 9), (2, 40)):
        feats = torch.from_numpy(rng.random((B, 1, 229, T)).astype(np.float32))
        with torch.no_grad():
            ref = model_ref.run_on_batch(feats, sdt)
            got = model.engine_logits(feats.cuda())
        for key in ('onsets', 'multi_pitch'):
            assert (torch.sigmoid(got[key].cpu()) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4
            assert (got[key].cpu() - ref['logits'][key]).abs().max().item() < 1.5e-4


@pytest.mark.parametrize('dim_in,in_channels', [(229, 1), (72, 6)])
def test_complexity_3_engine_vs_oracle_on_ragged_batches(dim_in, in_channels):
    """OnsetsFrames (no offset head) at model_complexity 3 against the oracle on fresh inputs, batch and frame counts that do not
    fill the kernels' tiles; one-channel mel features and the six-channel HCQT shape (two K steps in the fused first conv)."""
    from oracle import model_ref
    from amt_tools_amd.models import OnsetsFrames
    sd = synth_state_dict(11, dim_in=dim_in, in_channels=in_channels, model_complexity=3)
    model = OnsetsFrames(dim_in, tools.PianoProfile(), in_channels, 3, device='cuda:0', precision='x3')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    sdt = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(6)
    for B, T in ((1, 1), (3, 17), (17, 9), (2, 40)):
        feats = torch.from_numpy(rng.random((B, in_channels, dim_in, T)).astype(np.float32))
        with torch.no_grad():
            ref = model_ref.run_on_batch(feats, sdt)
            got = model.engine_logits(feats.cuda())
        for key in ('onsets', 'multi_pitch'):
            assert (torch.sigmoid(got[key].cpu()) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4
            assert (got[key].cpu() - ref['logits'][key]).abs().max().item() < 1.5e-4


@pytest.mark.parametrize('dim_in', [229, 40, 88, 54, 192, 8])
def test_complexity_2_engine_vs_oracle_feature_sizes_and_ragged_batches(dim_in):
    """OnsetsFrames at model_complexity 2 (the headline engine: Toeplitz first conv fused into conv2) against the oracle on fresh
    inputs: feature sizes whose tile width is / is not a multiple of the first conv's 4-column units and that need 1 .. 5
    frequency tiles, frame counts that do not fill the 16-frame tiles (halo-row units, padded rows), batches that do not fill
    the persistent grid.  x3 (the parity gate): logits within 1.5e-4 / activations within 1e-4; bf16: a loose 1e-1 screen for gross errors."""
    from oracle import model_ref
    from amt_tools_amd.models import OnsetsFrames
    sd = synth_state_dict(17, dim_in=dim_in, in_channels=1, model_complexity=2)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    rng = np.random.default_rng(dim_in)
    cases = ((1, 1), (3, 17), (17, 9), (2, 40), (1, 33))
    feats = [torch.from_numpy(rng.random((B, 1, dim_in, T)).astype(np.float32)) for B, T in cases]
    with torch.no_grad():
        refs = [model_ref.run_on_batch(f, sdt) for f in feats]
    for precision, tol in (('x3', 1.5e-4), ('bf16', 1e-1)):
        model = OnsetsFrames(dim_in, tools.PianoProfile(), 1, 2, device='cuda:0', precision=precision)
        model.load_state_dict(sdt)
        model.change_device()
        model.eval()
        for f, ref in zip(feats, refs):
            with torch.no_grad():
                got = model.engine_logits(f.cuda())
            for key in ('onsets', 'multi_pitch', 'pitch_head'):
                err = (got[key].cpu() - ref['logits'][key]).abs().max().item()
                assert err < tol, (precision, tuple(f.shape), key, err)
                if precision == 'x3':
                    assert (torch.sigmoid(got[key].cpu()) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
@pytest.mark.parametrize('cls,B,T', [('OnsetsFrames', 523, 33), ('OnsetsFrames', 1024, 40), ('OnsetsFrames2', 261, 33)])
def test_engine_at_batch_sizes_that_take_the_eight_clip_recurrence(cls, B, T, precision):
    """DESIGN rule: every batch-size-dependent dispatch has a test on each side of its threshold.  lstm.hip switches to eight clips per
    block (bilstm4_kernel<.., NC = 2>) once four-clip blocks would outnumber the 256 CUs: more than 512 clips with one recurrent head,
    more than 256 with the two grouped recurrences of OnsetsFrames2 -- the mapping the headline bench (1024 clips) runs.  Whole engine
    against the CPU oracle for clips of the first block, blocks in the middle and the ragged last block (523 = 65 x 8 + 3, 261 = 32 x 8
    + 5), every clip distinct; x3: logits within 1.5e-4 and activations within 1e-4 (the parity gate), bf16: the 6e-2 logit bound."""
    from oracle import model_ref
    import amt_tools_amd.models as M
    offsets = cls == 'OnsetsFrames2'
    sd = synth_state_dict(23, dim_in=229, in_channels=1, model_complexity=2, offsets=offsets)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = getattr(M, cls)(229, tools.PianoProfile(), 1, 2, device='cuda:0', precision=precision)
    model.load_state_dict(sdt)
    model.change_device()
    model.eval()
    rng = np.random.default_rng(B + T)
    feats = torch.from_numpy(rng.random((B, 1, 229, T)).astype(np.float32))
    pick = sorted({0, 1, 7, 8, 9, B // 2, B // 2 + 1, B // 2 + 5, B - 9, B - 8, B - 4, B - 3, B - 2, B - 1})
    with torch.no_grad():
        got = model.engine_logits(feats.cuda())
        ref = model_ref.run_on_batch(feats[pick], sdt)
    tol = 1.5e-4 if precision == 'x3' else TOL['bf16']
    keys = ('onsets', 'multi_pitch', 'pitch_head') + (('offsets',) if offsets else ())
    for key in keys:
        g = got[key].cpu()[pick]
        err = (g - ref['logits'][key]).abs().max().item()
        assert err < tol, (key, err)
        if precision == 'x3':
            assert (torch.sigmoid(g) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4, key


# one-channel mel at model_complexity 2 (conv.hip / convf.hip), OnsetsFrames2 at 2 and at its default 3 (convg.hip, hidden-256 recurrence), HCQT
# (six input channels: convg.hip's fused first conv)
@pytest.mark.parametrize('name', ['of1_eval.npz', 'of2_mc2_eval.npz', 'of2_eval.npz', 'of1_hcqt_eval.npz'])
def test_f16_precision_matches_reference_golden(name):
    """precision 'f16': the bf16 engine with IEEE half operands (the second, -DAMTX_F16 build of conv / convf / gemm / lstm.hip: same
    matrix rate, three more mantissa bits).  Logits within 1e-2 of the REAL reference classes' outputs (bf16: 6e-2, x3: 1e-4) and several
    times closer to them than the bf16 mode's on the same input; piano rolls identical outside the tolerance band."""
    import amt_tools_amd.models as M
    g = load_golden(name)
    cls = M.OnsetsFrames2 if name.startswith('of2') else M.OnsetsFrames
    sd = synth_state_dict(int(g['seed']), dim_in=int(g['dim_in']), in_channels=int(g['in_channels']), model_complexity=int(g['model_complexity']),
                          offsets=name.startswith('of2'))
    errs = {}
    for precision in ('f16', 'bf16'):
        model = cls(int(g['dim_in']), tools.PianoProfile(), int(g['in_channels']), int(g['model_complexity']), device='cuda:0', precision=precision)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.change_device()
        model.eval()
        with torch.no_grad():
            out = model.run_on_batch({tools.KEY_FEATS: torch.from_numpy(g['feats'])})
            logits = model.engine_logits(torch.from_numpy(g['feats']).cuda())
        errs[precision] = max(np.abs(logits[key].cpu().numpy() - g['logits_' + key]).max() for key in ('onsets', 'multi_pitch', 'pitch_head'))
        if precision == 'f16':
            for key in ('onsets', 'multi_pitch'):
                near = np.abs(np.swapaxes(g['logits_' + key], -1, -2)) < 1e-2
                assert np.all((out[key].cpu().numpy() == g['out_' + key]) | near)
    assert errs['f16'] < 1e-2, errs
    assert errs['f16'] < 0.4 * errs['bf16'], errs


@pytest.mark.parametrize('dim_in', [229, 40, 8])
def test_f16_precision_vs_oracle_on_both_convolution_paths(dim_in):
    """f16 on the two-kernel path (small batches) and on the fused stack (256 or more head x clip x strip), ragged shapes; against the CPU
    oracle within 1e-2 (logits) / 2.5e-3 (activations)."""
    from oracle import model_ref
    import amt_tools_amd.models as M
    sd = synth_state_dict(19, dim_in=dim_in, in_channels=1, model_complexity=2)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = M.OnsetsFrames(dim_in, tools.PianoProfile(), 1, 2, device='cuda:0', precision='f16')
    model.load_state_dict(sdt)
    model.change_device()
    model.eval()
    rng = np.random.default_rng(dim_in + 1)
    for B, T in ((3, 17), (1, 33), (130, 47), (44, 140)):
        feats = torch.from_numpy(rng.random((B, 1, dim_in, T)).astype(np.float32))
        pick = sorted({0, B // 2, B - 1})
        with torch.no_grad():
            got = model.engine_logits(feats.cuda())
            ref = model_ref.run_on_batch(feats[pick], sdt)
        assert model._get_engine(torch.device('cuda:0')).conv_stack_fused(B, T) == (2 * B * ((T + 61) // 62) >= 256)
        for key in ('onsets', 'multi_pitch', 'pitch_head'):
            gk = got[key].cpu()[pick]
            assert (gk - ref['logits'][key]).abs().max().item() < 1e-2, (B, T, key)
            assert (torch.sigmoid(gk) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 2.5e-3, (B, T, key)


@pytest.mark.parametrize('precision', ['bf16', 'x3', 'f16'])
@pytest.mark.parametrize('cls,mc,dim_in,ic', [('OnsetsFrames', 2, 229, 1), ('OnsetsFrames2', 2, 229, 1), ('OnsetsFrames2', 3, 229, 1), ('OnsetsFrames', 2, 72, 6),
                                              ('OnsetsFrames', 3, 72, 6), ('OnsetsFrames', 4, 229, 1), ('OnsetsFrames2', 4, 72, 3), ('OnsetsFrames', 5, 229, 1),
                                              ('OnsetsFrames2', 5, 72, 3)])
def test_device_side_weight_sync_equals_the_host_path(cls, mc, dim_in, ic, precision, monkeypatch):
    """A weight RE-sync packs on the GPU (pack.hip, amtx_of_model_finalize_device) with the host packers' arithmetic: after the same
    parameter update, an engine re-synced on the device and one re-synced through the host (AMTX_HOST_WEIGHT_SYNC=1) return identical
    bits -- BatchNorm statistics, every convolution / Linear / LSTM tensor and the fp64-folded pitch head included.  Every engine
    configuration: conv.hip / convf.hip fragments (model_complexity 2, one channel), convg.hip's chunked fragments with and without a
    16-channel tail and its fused first conv (3, 4, multi-channel input), hidden 128 / 256 / 384 recurrences, the unfused first conv
    (x3 at model_complexity 4)."""
    import amt_tools_amd.models as M
    offsets = cls == 'OnsetsFrames2'
    sd = synth_state_dict(9, dim_in=dim_in, in_channels=ic, model_complexity=mc, offsets=offsets)
    feats = torch.from_numpy(np.random.default_rng(4).random((3, ic, dim_in, 40)).astype(np.float32)).cuda()
    outs = {}
    for mode in ('device', 'host'):
        if mode == 'host':
            monkeypatch.setenv('AMTX_HOST_WEIGHT_SYNC', '1')
        else:
            monkeypatch.delenv('AMTX_HOST_WEIGHT_SYNC', raising=False)
        model = getattr(M, cls)(dim_in, tools.PianoProfile(), ic, mc, device='cuda:0', precision=precision)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.change_device()
        model.eval()
        with torch.no_grad():
            first = model.engine_logits(feats)                       # the first sync always goes through the host
            g = torch.Generator(device='cuda').manual_seed(1)
            for p_ in model.parameters():                            # "an optimizer step"
                p_.add_(torch.randn(p_.shape, generator=g, device='cuda') * 0.01 * p_.abs().mean())
            for b_ in model.buffers():
                if b_.dtype.is_floating_point:
                    b_.mul_(1.05)
            eng = model._get_engine(feats.device)
            assert eng.device_sync == (mode == 'device')
            got = model.engine_logits(feats)
            assert eng.device_sync == (mode == 'device') and eng.device_syncs == (1 if mode == 'device' else 0)   # the device path really ran
        assert not torch.equal(first['multi_pitch'], got['multi_pitch'])
        outs[mode] = got
    for k in outs['host']:
        assert torch.equal(outs['device'][k], outs['host'][k]), (k, (outs['device'][k] - outs['host'][k]).abs().max().item())


def test_device_side_weight_sync_falls_back_where_it_is_not_built(monkeypatch):
    """A multi-channel first conv on conv.hip's kernel (AMTX_NO_CONVG_MC2: the A/B switch back from convg.hip's) has no device packer:
    amtx_of_model_finalize_device answers AMTX_ERR_UNSUPPORTED, the engine notes it, re-syncs through the host and the results follow the
    new weights all the same."""
    import amt_tools_amd.models as M
    monkeypatch.setenv('AMTX_NO_CONVG_MC2', '1')
    model = M.OnsetsFrames(72, tools.PianoProfile(), 6, 2, device='cuda:0', precision='bf16')
    model.change_device()
    model.eval()
    feats = torch.rand(2, 6, 72, 20, device='cuda')
    with torch.no_grad():
        a = model.engine_logits(feats)['multi_pitch'].clone()
        for p_ in model.parameters():
            p_.mul_(1.1)
        b = model.engine_logits(feats)['multi_pitch']
    assert not model._get_engine(feats.device).device_sync
    assert not torch.equal(a, b)


def _of1_bf16(seed, dim_in, cls='OnsetsFrames'):
    import amt_tools_amd.models as M
    sd = synth_state_dict(seed, dim_in=dim_in, in_channels=1, model_complexity=2, offsets=cls == 'OnsetsFrames2')
    model = getattr(M, cls)(dim_in, tools.PianoProfile(), 1, 2, device='cuda:0', precision='bf16')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    return model


@pytest.mark.parametrize('dim_in', [229, 40, 88, 54, 192, 8, 5])
def test_fused_conv_stack_is_bit_identical_to_the_two_kernel_path(dim_in, monkeypatch):
    """convf.hip (layer1 -> layer2 -> layer3 in one kernel, both intermediate maps in LDS only) accumulates every output in the order
    conv.hip's two kernels do, so the engine's logits must be the SAME BITS with and without it (AMTX_NO_CONV_FUSE=1 is read when the
    engine is created).  Feature sizes that need 1 .. 15 frequency steps and end inside a step, frame counts that end inside the first /
    second / third 16-row tile of a 62-frame strip, one and several strips per clip, two and three heads; batch sizes on BOTH sides of the
    dispatch threshold (heads x clips x strips >= 256: DESIGN rule), checked through the workspace size (the 32-channel map behind layer2
    has no HBM buffer on the fused path)."""
    from amt_tools_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(dim_in)
    cases = [('OnsetsFrames', 128, 33), ('OnsetsFrames', 127, 33), ('OnsetsFrames', 43, 140), ('OnsetsFrames', 130, 47),
             ('OnsetsFrames', 260, 1), ('OnsetsFrames', 129, 17), ('OnsetsFrames2', 86, 46), ('OnsetsFrames2', 44, 93),
             # frame counts on both sides of the 62-frame strip boundaries (one strip exactly full, one frame into the next, two strips, ...)
             ('OnsetsFrames', 130, 62), ('OnsetsFrames', 66, 63), ('OnsetsFrames', 65, 124), ('OnsetsFrames', 44, 125), ('OnsetsFrames', 33, 187),
             # strips of 60 frames without the ninth layer1 unit where 62 would not save a strip (T = 60, 120), 62 where it does (61, 121)
             ('OnsetsFrames', 130, 60), ('OnsetsFrames', 130, 61), ('OnsetsFrames', 65, 120), ('OnsetsFrames', 65, 121)]
    if dim_in == 229:
        cases.append(('OnsetsFrames', 12, 625))             # the BASELINE clip length: eleven strips, the last one five frames long
    for cls, B, T in cases:
        feats = torch.from_numpy(rng.random((B, 1, dim_in, T)).astype(np.float32)).cuda()
        got = {}
        for mode in ('fused', 'two-kernel'):
            if mode == 'two-kernel':
                monkeypatch.setenv('AMTX_NO_CONV_FUSE', '1')
            else:
                monkeypatch.delenv('AMTX_NO_CONV_FUSE', raising=False)
            model = _of1_bf16(31, dim_in, cls)
            with torch.no_grad():
                got[mode] = {k: v.clone() for k, v in model.engine_logits(feats).items()}
            got[mode + '_ws'] = L.amtx_of_workspace_bytes(model._get_engine(feats.device).handle, B, T)
            got[mode + '_q'] = model._get_engine(feats.device).conv_stack_fused(B, T)
            del model
        heads = 3 if cls == 'OnsetsFrames2' else 2
        expect_fused = heads * B * ((T + 61) // 62) >= 256
        assert (got['fused_ws'] < got['two-kernel_ws']) == expect_fused, (cls, B, T, got['fused_ws'], got['two-kernel_ws'])
        assert got['fused_q'] == expect_fused and not got['two-kernel_q']
        for k in got['fused']:
            assert torch.equal(got['fused'][k], got['two-kernel'][k]), (cls, B, T, k, (got['fused'][k] - got['two-kernel'][k]).abs().max().item())


def test_whole_tracks_of_several_thousand_frames(monkeypatch):
    """The reference transcribes WHOLE tracks in one forward pass (transcribe.py / evaluate.py feed run_on_batch a track's full feature
    matrix: thousands of frames, batch 1 - 2).  2 tracks x 4001 frames (65 strips per track, the last one 33 frames long; 4001 dependent
    recurrence steps): the fused stack (2 heads x 2 x 65 = 260 strips; its output in planes per frequency column or, AMTX_OF_ROWMAJOR_A3=1,
    row-major) returns the bits of the two-kernel path, and the x3 engine stays
    within 1.5e-4 of the CPU oracle at the start, in the middle and at the end of the track."""
    from oracle import model_ref
    import amt_tools_amd.models as M
    B, T, dim_in = 2, 4001, 229
    feats = torch.from_numpy(np.random.default_rng(8).random((B, 1, dim_in, T)).astype(np.float32))
    got = {}
    for mode in ('fused', 'two-kernel'):
        if mode == 'two-kernel':
            monkeypatch.setenv('AMTX_NO_CONV_FUSE', '1')
        else:
            monkeypatch.delenv('AMTX_NO_CONV_FUSE', raising=False)
        model = _of1_bf16(32, dim_in)
        assert model._get_engine(torch.device('cuda:0')).conv_stack_fused(B, T) == (mode == 'fused')
        with torch.no_grad():
            got[mode] = {k: v.clone() for k, v in model.engine_logits(feats.cuda()).items()}
        del model
    for k in got['fused']:
        assert torch.equal(got['fused'][k], got['two-kernel'][k]), k
    # (the A/B switch is read once per process: the row-major layout of the fused stack's output is compared in a fresh interpreter)
    import subprocess, sys, os
    code = ("import numpy as np, torch, sys; sys.path.insert(0, 'tests'); import test_gpu_model as t; "
            "f = torch.from_numpy(np.random.default_rng(8).random((2, 1, 229, 4001)).astype(np.float32)).cuda(); "
            "m = t._of1_bf16(32, 229); assert m._get_engine(torch.device('cuda:0')).conv_stack_fused(2, 4001); "
            "torch.save({k: v.cpu() for k, v in m.engine_logits(f).items()}, sys.argv[1])")
    out_path = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'amtx_rowmajor_a3_{os.getpid()}.pt')
    env = dict(os.environ, AMTX_OF_ROWMAJOR_A3='1')
    env.pop('AMTX_NO_CONV_FUSE', None)
    subprocess.run([sys.executable, '-c', code, out_path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    rowmajor = torch.load(out_path)
    os.remove(out_path)
    for k in got['fused']:
        assert torch.equal(got['fused'][k].cpu(), rowmajor[k]), k
    monkeypatch.delenv('AMTX_NO_CONV_FUSE', raising=False)
    sd = synth_state_dict(32, dim_in=dim_in, in_channels=1, model_complexity=2)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = M.OnsetsFrames(dim_in, tools.PianoProfile(), 1, 2, device='cuda:0', precision='x3')
    model.load_state_dict(sdt)
    model.change_device()
    model.eval()
    torch.set_num_threads(8)
    with torch.no_grad():
        x3 = model.engine_logits(feats.cuda())
        ref = model_ref.run_on_batch(feats[:1], sdt)          # the recurrences make every frame depend on the whole track: no cropping
    for key in ('onsets', 'multi_pitch', 'pitch_head'):
        for lo, hi in ((0, 64), (1970, 2034), (T - 64, T)):
            err = (x3[key][0, lo:hi].cpu() - ref['logits'][key][0, lo:hi]).abs().max().item()
            assert err < 1.5e-4, (key, lo, err)
        assert (got['fused'][key][0].cpu() - ref['logits'][key][0]).abs().max().item() < 6e-2, key


def test_fused_conv_stack_on_raw_power_features_is_bit_identical(monkeypatch):
    """The same, entered through amtx_of_forward_power (audio -> log-mel power -> engine, dB scaling applied while the features are staged):
    piano rolls of run_on_batch on audio with and without the fused stack."""
    from amt_tools_amd.features import MelSpec
    mod = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)
    audio = torch.from_numpy(np.stack([synth_clip(i, num_samples=512 * 60) for i in range(140)])).cuda()
    outs = []
    for mode in ('fused', 'two-kernel'):
        if mode == 'two-kernel':
            monkeypatch.setenv('AMTX_NO_CONV_FUSE', '1')
        else:
            monkeypatch.delenv('AMTX_NO_CONV_FUSE', raising=False)
        model = _of1_bf16(3, 229)
        model.frontend = torch.nn.Sequential(mod.frontend())
        model.change_device()
        model.eval()
        with torch.no_grad():
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
        outs.append({k: out[k].clone() for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH)})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert outs[0][tools.KEY_ONSETS].shape == (140, 88, 61)


@pytest.mark.parametrize('mc', [2, 3])
@pytest.mark.parametrize('precision', ['bf16', 'x3'])
def test_engine_is_deterministic_run_to_run(mc, precision):
    """Same input, same workspace contents or not: identical bits.  (The first general-channel conv kernel mixed the legacy 16-deep
    bf16 MFMA into 32-deep accumulation chains and produced run-to-run varying values; this is the screen for that class of bug.)"""
    from amt_tools_amd.models import OnsetsFrames2
    sd = synth_state_dict(5, dim_in=229, in_channels=1, model_complexity=mc, offsets=True)
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, mc, device='cuda:0', precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    feats = torch.from_numpy(np.random.default_rng(2).random((3, 1, 229, 70)).astype(np.float32)).cuda()
    with torch.no_grad():
        first = {k: v.clone() for k, v in model.engine_logits(feats).items()}
        model._get_engine(feats.device).workspace.fill_(0xFF)      # stale workspace contents (here: NaN bit patterns) must not matter
        for _ in range(3):
            again = model.engine_logits(feats)
            for k in first:
                assert torch.equal(first[k], again[k]), k


def test_unbuilt_model_complexity_is_rejected_loudly_by_the_engine():
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd._lib import AmtxError
    model = OnsetsFrames(229, tools.PianoProfile(), 1, 6, device='cuda:0')        # 96 / 96 / 192 channels, hidden 640: not built
    model.change_device()
    model.eval()
    with pytest.raises(AmtxError), torch.no_grad():
        model.run_on_batch({tools.KEY_FEATS: torch.zeros(1, 1, 229, 8)})


def test_forward_power_is_refused_loudly_where_the_conv_kernel_does_not_stage_features():
    """amtx_of_forward_power exists for one-channel models whose first conv is fused into conv.hip's kernel; model_complexity 3 (convg.hip)
    and multi-channel inputs answer amtx_of_fuses_db_scale() = 0, run_on_batch takes the ordinary feature path for them, and a direct
    call with pending power features raises instead of computing on unscaled values."""
    from amt_tools_amd._lib import AmtxError
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.models import OnsetsFrames, PendingFeatures
    mod = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)
    audio = torch.from_numpy(np.stack([synth_clip(i, num_samples=512 * 20) for i in range(2)])).cuda()
    for mc, fuses in ((2, True), (3, False)):
        sd = synth_state_dict(3, dim_in=229, in_channels=1, model_complexity=mc)
        model = OnsetsFrames(229, tools.PianoProfile(), 1, mc, device='cuda:0', precision='bf16')
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.frontend = torch.nn.Sequential(mod.frontend())
        model.change_device()
        model.eval()
        eng = model._get_engine(torch.device('cuda:0'))
        assert eng.fuses_db_scale() == fuses
        with torch.no_grad():
            out = model.run_on_batch({tools.KEY_AUDIO: audio})                      # both complexities: fine through run_on_batch
            assert out[tools.KEY_ONSETS].shape == (2, 88, 21)
            power, cmax = mod.power_batch(audio)
            if not fuses:
                with pytest.raises(AmtxError):
                    model(PendingFeatures(mod, power, cmax))


def test_engine_ragged_batch_and_single_frame():
    from oracle import model_ref
    g = load_golden('of1_eval.npz')
    model = _model(g, 'x3')
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(5)
    for B, T in ((1, 1), (3, 17), (17, 9)):
        feats = torch.from_numpy(rng.random((B, 1, 229, T)).astype(np.float32))
        with torch.no_grad():
            ref = model_ref.run_on_batch(feats, sd)
            got = model.engine_logits(feats.cuda())
        for key in ('onsets', 'multi_pitch'):
            # pre-threshold piano-roll activations (sigmoid) within 1e-4; raw logits (|x| up to ~5) within 1.5e-4
            assert (torch.sigmoid(got[key].cpu()) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4
            assert (got[key].cpu() - ref['logits'][key]).abs().max().item() < 1.5e-4


def test_weights_are_resynced_after_an_update_and_model_pickles():
    import io
    g = load_golden('of1_eval.npz')
    model = _model(g, 'x3')
    feats = torch.from_numpy(g['feats']).cuda()
    with torch.no_grad():
        a = model.engine_logits(feats)['onsets'].clone()
        model.onset_head[2].output_layer.bias.add_(1.0)
        b = model.engine_logits(feats)['onsets'].clone()
    assert torch.allclose(b, a + 1.0, atol=1e-5)
    buf = io.BytesIO()
    torch.save(model, buf)                                              # train.py:172 pickles the whole module
    buf.seek(0)
    clone = torch.load(buf, map_location='cuda:0', weights_only=False)
    clone.change_device('cuda:0')
    clone.eval()
    with torch.no_grad():
        c = clone.engine_logits(feats)['onsets']
    assert torch.allclose(c, b, atol=1e-6)


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
def test_fused_frontend_audio_to_pianoroll(precision):
    """audio -> (HIP mel front-end as model.frontend) -> engine, against oracle front-end + oracle model."""
    from oracle import frontend_np as fe, model_ref
    from amt_tools_amd.features import MelSpec
    g = load_golden('of1_eval.npz')
    model = _model(g, precision)
    mod = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)
    model.frontend = torch.nn.Sequential(mod.frontend())
    audio = np.stack([synth_clip(i, num_samples=512 * 24 - 1) for i in range(2)])
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_AUDIO: torch.from_numpy(audio)})
    feats = np.stack([fe.melspec_process_audio(a, 22050) for a in audio]).astype(np.float32)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith('frontend')}
    with torch.no_grad():
        ref = model_ref.run_on_batch(torch.from_numpy(feats), sd)
    tol = TOL[precision] * 2
    for key in ('onsets', 'multi_pitch'):
        near = np.abs(ref['logits'][key].transpose(-1, -2).numpy()) < tol
        assert np.all((out[key].cpu().numpy() == ref[key].numpy()) | near)
        assert out[key].shape == (2, 88, 24)


@pytest.mark.parametrize('precision', ['x3', 'bf16'])
@pytest.mark.parametrize('frontend', ['mel', 'stft'])
def test_db_scaling_deferred_into_the_conv_kernel_is_bit_identical(frontend, precision):
    """run_on_batch on raw audio defers the dB scaling into the first conv kernel (amtx_of_forward_power): the logits must be the
    SAME BITS as for the feature tensor the front-end would have written (amtx_spec_scale + amtx_of_forward), also for clips of
    very different level (per-clip reference and top_db floor), a silent clip and a ragged last tile."""
    from amt_tools_amd.features import MelSpec, STFT
    from amt_tools_amd.models import OnsetsFrames, PendingFeatures
    if frontend == 'mel':
        mod, dim_in = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048), 229
    else:
        mod, dim_in = STFT(sample_rate=16000, hop_length=512, n_fft=256), 129
    sd = synth_state_dict(5, dim_in=dim_in, in_channels=1, model_complexity=2)
    model = OnsetsFrames(dim_in, tools.PianoProfile(), 1, 2, device='cuda:0', precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.frontend = torch.nn.Sequential(mod.frontend())
    model.change_device()
    model.eval()
    audio = np.stack([synth_clip(i, num_samples=512 * 37 + 11) for i in range(4)])
    audio[1] *= 1e-3
    audio[2] *= 30.0
    audio[3] = 0.0
    a = torch.from_numpy(audio).cuda()
    with torch.no_grad():
        batch = {tools.KEY_AUDIO: a}
        model.__dict__['_in_run_on_batch'] = True
        try:
            pre = model.pre_proc(batch)
        finally:
            model.__dict__.pop('_in_run_on_batch')
        assert isinstance(pre[tools.KEY_FEATS], PendingFeatures)                     # the deferred path is the one that runs
        assert torch.is_tensor(model.pre_proc(batch)[tools.KEY_FEATS])               # pre_proc on its own still returns features
        fused = model(pre[tools.KEY_FEATS])
        feats = pre[tools.KEY_FEATS].materialize()
        np.testing.assert_array_equal(feats.cpu().numpy(), model.pre_proc(batch)[tools.KEY_FEATS].cpu().numpy())
        plain = model(feats)
        out = model.run_on_batch(batch)
        # a caller-supplied reference power per clip (track-level dB reference) goes through the same kernel argument
        pf = pre[tools.KEY_FEATS]
        with_ref = PendingFeatures(pf.module, pf.power, pf.clip_max, ref=(pf.clip_max * 3.0 + 1e-3).contiguous())
        fused_ref, plain_ref = model(with_ref), model(with_ref.materialize())
    for key in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        np.testing.assert_array_equal(fused[key].cpu().numpy(), plain[key].cpu().numpy())
        np.testing.assert_array_equal(fused_ref[key].cpu().numpy(), plain_ref[key].cpu().numpy())
        assert not np.array_equal(fused_ref[key].cpu().numpy(), fused[key].cpu().numpy())      # the reference really changes the features
        assert out[key].shape == (4, 88, 38)
        np.testing.assert_array_equal(out[key].cpu().numpy(), (plain[key].transpose(-1, -2) > 0).float().cpu().numpy())


def test_of2_experiment_shape_audio_to_notes():
    """The reference's OnsetsFrames2 experiment end to end (scripts of_2.py:87-110,157-175): audio -> HTK log-mel front-end fused
    into the model -> OnsetsFrames2 as shipped (model_complexity 3, offset head) -> piano rolls + offset probabilities -> notes.
    Against oracle front-end + oracle model (fp32-class mode), and the batched driver against one-clip-at-a-time run_offline."""
    from oracle import frontend_np as fe, model_ref
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.inference import run_offline, run_offline_batched
    from amt_tools_amd.models import OnsetsFrames2
    from amt_tools_amd.transcribe import NoteTranscriber
    sd = synth_state_dict(21, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, device='cuda:0', precision='x3')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    mod = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048, htk=True)
    model.frontend = torch.nn.Sequential(mod.frontend())
    T = 40
    clips = np.stack([synth_clip(i, num_samples=512 * T - 1) for i in range(5)])
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_AUDIO: torch.from_numpy(clips[:2])})
    assert set(out.keys()) == {tools.KEY_ONSETS, tools.KEY_OFFSETS, tools.KEY_MULTIPITCH}
    feats = np.stack([fe.melspec_process_audio(a, 22050, htk=True) for a in clips[:2]]).astype(np.float32)
    sdt = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith('frontend')}
    with torch.no_grad():
        ref = model_ref.run_on_batch(torch.from_numpy(feats), sdt)
    for key in ('onsets', 'multi_pitch'):
        near = np.abs(ref['logits'][key].transpose(-1, -2).numpy()) < 2e-4
        assert out[key].shape == (2, 88, T)
        assert np.all((out[key].cpu().numpy() == ref[key].numpy()) | near)
    off = out[tools.KEY_OFFSETS].cpu().numpy()
    assert off.shape == (2, 88, T) and off.min() >= 0.0 and off.max() <= 1.0            # probabilities (onsetsframes.py:323-325)
    assert np.abs(off - torch.sigmoid(ref['logits']['offsets']).transpose(-1, -2).numpy()).max() < 1e-4
    times = (np.arange(T) * 512 / 22050.0).astype(np.float32)
    res = run_offline_batched(clips, model, times=times, batch_size=2, decode_notes=True)
    est = NoteTranscriber(tools.PianoProfile())
    for i in (0, 4):
        single = run_offline({tools.KEY_AUDIO: clips[i], tools.KEY_TIMES: times}, model, est)
        for key in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH, tools.KEY_OFFSETS, tools.KEY_NOTES):
            np.testing.assert_array_equal(res[i][key], single[key])


def test_full_size_batch_properties():
    """BASELINE-size clips (625 frames): clip independence (a clip's result does not depend on its batch
    neighbours) and time-reversal consistency are size-independent properties of the path."""
    g = load_golden('of1_eval.npz')
    model = _model(g, 'bf16')
    rng = np.random.default_rng(9)
    feats = torch.from_numpy(rng.random((5, 1, 229, 625)).astype(np.float32)).cuda()
    with torch.no_grad():
        full = model.engine_logits(feats)
        solo = model.engine_logits(feats[3:4])
    for key in ('onsets', 'multi_pitch'):
        assert torch.equal(full[key][3], solo[key][0])
        assert torch.isfinite(full[key]).all()


@pytest.mark.parametrize('precision', ['bf16', 'x3'])
def test_full_size_batch_properties_complexity_3(precision):
    """The same size-independent property for the general-channel conv kernels / streaming recurrences of model_complexity 3 at
    the BASELINE clip length: a clip's logits do not depend on its batch neighbours (persistent blocks walk tiles of several
    clips; the recurrence packs 16 clips per block), and reversing the batch order permutes the results."""
    from amt_tools_amd.models import OnsetsFrames2
    sd = synth_state_dict(41, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, device='cuda:0', precision=precision)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    rng = np.random.default_rng(10)
    feats = torch.from_numpy(rng.random((19, 1, 229, 625)).astype(np.float32)).cuda()
    with torch.no_grad():
        full = {k: v.clone() for k, v in model.engine_logits(feats).items()}
        solo = {k: v.clone() for k, v in model.engine_logits(feats[17:18]).items()}
        flipped = model.engine_logits(feats.flip(0))
    for key in ('onsets', 'offsets', 'multi_pitch'):
        assert torch.equal(full[key][17], solo[key][0]), key
        assert torch.equal(full[key], flipped[key].flip(0)), key
        assert torch.isfinite(full[key]).all()


def test_config3_hcqt_frontend_fused_into_the_model():
    """BASELINE config 3: audio -> HIP HCQT (6 x 72) as model.frontend -> OnsetsFrames(dim_in=72, in_channels=6)."""
    from oracle import cqt_np as cq, model_ref
    from amt_tools_amd.features import HCQT
    g = load_golden('of1_hcqt_eval.npz')
    model = _model(g, 'x3')
    mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    model.frontend = torch.nn.Sequential(mod.frontend())
    audio = np.stack([synth_clip(i, num_samples=512 * 30 - 1) for i in range(2)])
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_AUDIO: torch.from_numpy(audio)})
    feats = np.stack([cq.hcqt_process_audio(a, n_bins=72) for a in audio]).astype(np.float32)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith('frontend')}
    with torch.no_grad():
        ref = model_ref.run_on_batch(torch.from_numpy(feats), sd)
    for key in ('onsets', 'multi_pitch'):
        assert out[key].shape == ref[key].shape == (2, 88, feats.shape[-1])
        near = np.abs(ref['logits'][key].transpose(-1, -2).numpy()) < 5e-3      # CQT feature tolerance (1e-3) through the model
        assert np.all((out[key].cpu().numpy() == ref[key].numpy()) | near)
        assert (out[key].cpu().numpy() != ref[key].numpy()).mean() < 5e-3


@pytest.mark.timeout(900)
def test_config3_at_full_clip_length_matches_the_oracle(capsys):
    """BASELINE config 3 at its real size: 4 clips x 319 999 samples (625 frames) -- audio -> HIP HCQT (6 harmonics x
    72 bins, amt_tools/features/hvqt.py:107-133) -> OnsetsFrames(dim_in 72, 6 channels, mc 2) in the x3 precision -> piano rolls.
    (1) the HCQT map against oracle/cqt_np.hcqt_process_audio.  LINEAR magnitudes within 6e-6 of each harmonic's map maximum (measured
        1.4e-6 .. 3.9e-6: the two-plane bf16 products of the basis kernel carry 2^-17 per term) -- a fixed absolute error, which the dB
        function divides by the level: scaled features within 5e-4 for cells above -60 dB (>= 0.25; measured 2e-4) and within 4e-3
        between the -80 dB clamp and -60 dB (measured 2.6e-3 at -76 dB, where 3e-6 of the maximum IS 3 % of the value).  The 1e-3 of
        tests/test_gpu_cqt.py holds on its 30 000-sample clips; at 319 999 samples more cells sit just above the floor;
    (2) the engine on the GPU's OWN feature map against the model oracle on the same map: logits within 1.5e-4, activations within 1e-4
        (the x3 gate: isolates the model from the front-end's tolerance);
    (3) end to end (audio -> rolls through run_on_batch, HCQT fused as model.frontend) against oracle front-end -> oracle model: rolls
        identical except where the oracle's logit is inside the band the 1e-3 feature tolerance maps to; the band is measured here as the
        largest end-to-end logit difference and bounded."""
    from oracle import cqt_np as cq, model_ref
    from amt_tools_amd.features import HCQT
    g = load_golden('of1_hcqt_eval.npz')
    model = _model(g, 'x3')
    mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    model.frontend = torch.nn.Sequential(mod.frontend())
    audio = np.stack([synth_clip(70 + i) for i in range(4)])
    assert audio.shape == (4, 319999)
    x = torch.from_numpy(audio).cuda()
    with torch.no_grad():
        feats_gpu = mod.process_batch(x)                                            # (4, 6, 72, 625)
        out = model.run_on_batch({tools.KEY_AUDIO: torch.from_numpy(audio)})
        logits = model.engine_logits(feats_gpu)
    feats_ref = np.stack([cq.hcqt_process_audio(a, sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12) for a in audio]).astype(np.float32)
    assert feats_gpu.shape == feats_ref.shape == (4, 6, 72, 625)
    err_map = np.abs(feats_gpu.cpu().numpy() - feats_ref)
    err_fe, err_fe_hi = float(err_map.max()), float(err_map[feats_ref >= 0.25].max())
    assert err_fe_hi < 5e-4 and err_fe < 4e-3, (err_fe_hi, err_fe)
    lin = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, decibels=False)
    with torch.no_grad():
        lin_gpu = lin.process_batch(x[:2]).cpu().numpy()
    err_lin = 0.0
    for i in range(2):
        lin_ref = cq.hcqt_process_audio(audio[i], sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, decibels=False)
        for h in range(6):
            err_lin = max(err_lin, float(np.abs(lin_gpu[i, h] - lin_ref[h]).max() / lin_ref[h].max()))
    assert err_lin < 6e-6, err_lin
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith('frontend')}
    old_impl, model_ref.LSTM_IMPL = model_ref.LSTM_IMPL, 'aten'
    try:
        with torch.no_grad():
            ref_same = model_ref.run_on_batch(feats_gpu.cpu(), sd)                  # the model oracle on the GPU's own features
            ref = model_ref.run_on_batch(torch.from_numpy(feats_ref), sd)           # the whole oracle path
    finally:
        model_ref.LSTM_IMPL = old_impl
    err_model = err_act = err_e2e = 0.0
    for key in ('onsets', 'multi_pitch'):
        a, b = logits[key].cpu(), ref_same['logits'][key]
        err_model = max(err_model, float((a - b).abs().max()))
        err_act = max(err_act, float((torch.sigmoid(a) - torch.sigmoid(b)).abs().max()))
        err_e2e = max(err_e2e, float((a - ref['logits'][key]).abs().max()))
    assert err_model < 1.5e-4 and err_act < 1e-4, (err_model, err_act)
    assert err_e2e < 3e-3, err_e2e                             # measured 6.1e-4
    cells = diff = 0
    for key in ('onsets', 'multi_pitch'):
        got, want = out[key].cpu().numpy(), ref[key].numpy()
        assert got.shape == want.shape == (4, 88, 625)
        near = np.abs(ref['logits'][key].transpose(-1, -2).numpy()) <= err_e2e * 1.01 + 1e-6
        assert np.all((got == want) | near), key
        cells += got.size
        diff += int((got != want).sum())
        # and against the model oracle on the SAME features the rolls are the reference's outside the x3 band
        same = ref_same[key].numpy()
        near_s = np.abs(ref_same['logits'][key].transpose(-1, -2).numpy()) <= 1.5e-4
        assert np.all((got == same) | near_s), key
    assert diff / cells < 2e-3, diff / cells
    with capsys.disabled():
        print(f'\n[config 3 full size] 4 clips x 319999 samples: HCQT linear magnitudes within {err_lin:.2e} of the map maximum; scaled map max abs err {err_fe_hi:.2e} above -60 dB, '
              f'{err_fe:.2e} down to the -80 dB clamp; x3 engine vs oracle on the same map: logits '
              f'{err_model:.2e}, activations {err_act:.2e}; end to end: logits {err_e2e:.2e}, {diff} of {cells} piano-roll cells differ')


def test_run_offline_batched_pipeline_equals_per_clip_runs():
    """BASELINE config 5 driver: the three-stage pipeline (upload on a copy stream / kernels / host assembly of the previous
    batch) returns, per clip, exactly what one-clip-at-a-time `run_offline` + the host NoteTranscriber return; shards by rank
    partition the clips."""
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.inference import run_offline, run_offline_batched
    from amt_tools_amd.transcribe import NoteTranscriber
    g = load_golden('of1_eval.npz')
    model = _model(g, 'x3')
    model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048).frontend())
    clips = np.stack([synth_clip(i, num_samples=512 * 40 - 1) for i in range(7)])
    times = (np.arange(40) * 512 / 22050.0).astype(np.float32)      # run_offline casts every array of the track to float32
    res = run_offline_batched(clips, model, times=times, batch_size=3, decode_notes=True)          # batches of 3, 3, 1
    assert sorted(res) == list(range(7))
    est = NoteTranscriber(tools.PianoProfile())
    for i in (0, 3, 6):
        single = run_offline({tools.KEY_AUDIO: clips[i], tools.KEY_TIMES: times}, model, est)
        np.testing.assert_array_equal(res[i][tools.KEY_ONSETS], single[tools.KEY_ONSETS])
        np.testing.assert_array_equal(res[i][tools.KEY_MULTIPITCH], single[tools.KEY_MULTIPITCH])
        np.testing.assert_array_equal(res[i][tools.KEY_NOTES], single[tools.KEY_NOTES])
    shard = run_offline_batched(clips, model, times=times, batch_size=2, rank=1, world=2, decode_notes=True, keep=())
    assert sorted(shard) == [1, 3, 5] and set(shard[1]) == {tools.KEY_NOTES}
    np.testing.assert_array_equal(shard[3][tools.KEY_NOTES], res[3][tools.KEY_NOTES])


def test_run_offline_batched_takes_16_bit_pcm_and_returns_the_float32_results():
    """Round 6: int16 clips travel to the GPU as they are (half the PCIe bytes) and become sample / 32768 there -- exactly the float32 array the
    reference's loader makes of a 16-bit file (amt_tools/tools/io.py:80-82) -- so every result equals the float32 hand-over's, bit for bit;
    contiguous batches from pinned and pageable memory, a sharded (strided) selection, and the CPU model."""
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.inference import run_offline_batched
    g = load_golden('of1_eval.npz')
    model = _model(g, 'x3')
    model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048).frontend())
    f = np.stack([synth_clip(i, num_samples=512 * 40 - 1) for i in range(7)])
    pcm = np.clip(np.round(f / np.abs(f).max() * 32767.0), -32768, 32767).astype(np.int16)
    as_float = (pcm.astype(np.float32) / 32768.0).astype(np.float32)
    times = (np.arange(40) * 512 / 22050.0).astype(np.float32)
    ref = run_offline_batched(as_float, model, times=times, batch_size=3, decode_notes=True)
    for src in (pcm, torch.from_numpy(pcm).pin_memory()):
        got = run_offline_batched(src, model, times=times, batch_size=3, decode_notes=True)
        assert sorted(got) == sorted(ref)
        for i in ref:
            for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH, tools.KEY_NOTES):
                np.testing.assert_array_equal(got[i][k], ref[i][k])
    shard = run_offline_batched(pcm, model, times=times, batch_size=2, rank=1, world=2, decode_notes=True)
    assert sorted(shard) == [1, 3, 5]
    for i in shard:
        np.testing.assert_array_equal(shard[i][tools.KEY_NOTES], ref[i][tools.KEY_NOTES])


def test_engine_long_single_clip():
    """One long track (T = 1500 frames, not a multiple of the 16-frame conv tile, of 4, or of the reference's 512-frame LSTM
    chunk): the whole-sequence recurrence and the tiled convolutions against the oracle (fp32-class mode)."""
    from oracle import model_ref
    g = load_golden('of1_eval.npz')
    model = _model(g, 'x3')
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(11)
    feats = torch.from_numpy(rng.random((1, 1, 229, 1500)).astype(np.float32))
    with torch.no_grad():
        ref = model_ref.run_on_batch(feats, sd)
        got = model.engine_logits(feats.cuda())
    for key in ('onsets', 'multi_pitch'):
        assert (torch.sigmoid(got[key].cpu()) - torch.sigmoid(ref['logits'][key])).abs().max().item() < 1e-4
        assert (got[key].cpu() - ref['logits'][key]).abs().max().item() < 1.5e-4


def test_bench_workload_parity_of_both_precisions_against_the_cpu_oracle(capsys):
    """The headline workload itself (BASELINE config 2: audio -> HIP mel front-end -> engine -> piano rolls, 64 distinct synthetic
    clips x 625 frames, synthetic weights) in both engine precisions against the CPU oracle (numpy fp64 front-end + torch fp32
    model restatement).  SURVEY F8: on thresholded outputs the parity metric is the count of differing cells.
      x3  : logits within 2e-4 and activations (sigmoid) within 1e-4 of the oracle -- the mode that meets north_star's 1e-4 --
            and fewer than 1e-4 of the piano-roll cells differ (cells whose logit is ~0);
      bf16: the throughput mode; the fraction of differing cells is measured, printed and bounded at 2 x what MI355X measures
            (1.4e-3 of the cells, max |dlogit| 2.8e-2) -- and it is asserted to be OUTSIDE north_star's 1e-4, so that the log says which
            mode carries the parity claim (x3) and which one does not (bf16)."""
    from oracle import frontend_np as fe, model_ref
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.models import OnsetsFrames
    n_clips = 64
    sd = synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=2)
    sdt = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    clips = np.stack([synth_clip(i) for i in range(n_clips)])
    old_impl, model_ref.LSTM_IMPL = model_ref.LSTM_IMPL, 'aten'
    try:
        feats = np.stack([fe.melspec_process_audio(y, 22050, 512, 229, 2048, dtype=np.float32) for y in clips]).astype(np.float32)
        with torch.no_grad():
            ref = [model_ref.run_on_batch(torch.from_numpy(feats[i:i + 8]), sdt) for i in range(0, n_clips, 8)]
    finally:
        model_ref.LSTM_IMPL = old_impl
    ref_roll = {k: torch.cat([r[k] for r in ref]).numpy() for k in ('onsets', 'multi_pitch')}
    ref_logit = {k: torch.cat([r['logits'][k] for r in ref]) for k in ('onsets', 'multi_pitch')}
    audio = torch.from_numpy(clips).cuda()
    report = {}
    for precision in ('x3', 'bf16'):
        model = OnsetsFrames(229, tools.PianoProfile(), 1, 2, device='cuda:0', precision=precision)
        model.load_state_dict(sdt)
        model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048).frontend())
        model.change_device()
        model.eval()
        with torch.no_grad():
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
            logits = model.engine_logits(model.frontend(audio.unsqueeze(-2)))
        assert out[tools.KEY_ONSETS].shape == (n_clips, 88, 625)
        cells = diff = 0
        for k in ('onsets', 'multi_pitch'):
            got = out[k].cpu().numpy()
            cells += got.size
            diff += int((got != ref_roll[k]).sum())
        err_logit = max((logits[k].cpu() - ref_logit[k]).abs().max().item() for k in ref_logit)
        err_act = max((torch.sigmoid(logits[k].cpu()) - torch.sigmoid(ref_logit[k])).abs().max().item() for k in ref_logit)
        report[precision] = (diff / cells, err_logit, err_act)
    with capsys.disabled():
        for p, (rate, el, ea) in report.items():
            print(f'\n[parity, {n_clips} clips x 625 frames vs CPU oracle] {p}: cell mismatch rate {rate:.3e}, max |dlogit| {el:.3e}, max |dsigmoid| {ea:.3e}')
    rate, el, ea = report['x3']
    assert el < 2e-4 and ea < 1e-4 and rate < 1e-4, report['x3']
    rate, el, ea = report['bf16']
    assert rate < 3e-3 and el < 0.06, report['bf16']
    # status, made visible: bf16 operands do NOT meet the 1e-4 activation tolerance (measured 6.8e-3); only x3 does
    assert ea > 1e-4, ('bf16 is inside 1e-4 now: promote it to the parity mode and tighten its bounds', report['bf16'])
    with capsys.disabled():
        print(f'[parity] precision x3 is INSIDE the 1e-4 activation tolerance ({report["x3"][2]:.2e}); bf16 is OUTSIDE it ({ea:.2e})')


_EPILOGUE_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict
outs = {}
for name, cls, mc, T in (('of1', OnsetsFrames, 2, 70), ('of1_long', OnsetsFrames, 2, 333), ('of2', OnsetsFrames2, 2, 70)):
    sd = synth_state_dict(11, dim_in=229, in_channels=1, model_complexity=mc, offsets=(cls is OnsetsFrames2))
    model = cls(229, tools.PianoProfile(), 1, mc, device='cuda:0', precision='bf16')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    rng = np.random.default_rng(3)
    feats = torch.from_numpy(rng.random((5, 1, 229, T)).astype(np.float32))
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_FEATS: feats})                          # label-free: rolls only, logits stay in the engine
        lg = model.engine_logits(feats.cuda())                                        # logits wanted: every buffer is written
    for k, v in out.items():
        if torch.is_tensor(v):
            outs[f'{name}_roll_{k}'] = v.cpu().numpy()
    for k, v in lg.items():
        outs[f'{name}_logit_{k}'] = v.cpu().numpy()
np.savez(sys.argv[1], **outs)
'''


def test_head_gemm_epilogues_return_the_bits_of_the_separate_kernels(tmp_path):
    """bf16 mode writes the piano rolls and the refinement stage's bf16 input from the LogisticBank GEMMs' epilogues and skips buffers
    nobody reads; AMTX_OF_NO_ROLL_EPILOGUE=1 runs the separate pianoroll / conversion kernels.  Rolls (label-free run_on_batch),
    logits and offset probabilities (engine_logits) must be identical, for frame counts that put clip boundaries inside a 128-row
    GEMM tile, with and without the offset head."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('epilogue', {}), ('separate', {'AMTX_OF_NO_ROLL_EPILOGUE': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _EPILOGUE_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['epilogue']), np.load(files['separate'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 12
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert set(np.unique(a['of1_roll_' + tools.KEY_ONSETS])) <= {0.0, 1.0}


_X3_SPLIT_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict
outs = {}
for name, cls, B, T, F, C in (('of1', OnsetsFrames, 3, 70, 229, 1), ('of1_big', OnsetsFrames, 9, 333, 229, 1), ('of2', OnsetsFrames2, 2, 45, 229, 1),
                              ('hcqt', OnsetsFrames, 3, 70, 72, 6), ('hcqt_big', OnsetsFrames, 7, 401, 72, 6), ('ch3', OnsetsFrames, 2, 33, 40, 3)):
    sd = synth_state_dict(11, dim_in=F, in_channels=C, model_complexity=2, offsets=(cls is OnsetsFrames2))
    model = cls(F, tools.PianoProfile(), C, 2, device='cuda:0', precision='x3')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    rng = np.random.default_rng(3)
    feats = torch.from_numpy(rng.random((B, C, F, T)).astype(np.float32))
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_FEATS: feats})
        lg = model.engine_logits(feats.cuda())
    for k, v in out.items():
        if torch.is_tensor(v):
            outs[f'{name}_roll_{k}'] = v.cpu().numpy()
    for k, v in lg.items():
        outs[f'{name}_logit_{k}'] = v.cpu().numpy()
np.savez(sys.argv[1], **outs)
'''


def test_x3_split_plane_activations_return_the_bits_of_the_fp32_activation_path(tmp_path):
    """Round 5: in the x3 precision the dense layers' activations travel as two 16-bit planes written by the producing kernel's epilogue
    (conv2 -> conv3 -> fc1 -> input projection; AMTX_T_SPLIT) and the GEMMs DMA them into LDS; AMTX_X3_NO_SPLIT=1 keeps round 4's fp32
    activations, split by every consumer.  The planes are what the consumers computed themselves and the product order is the same, so
    logits, rolls and offset probabilities must be IDENTICAL -- for batches whose GEMM rows are below and above one 256-row tile, with and
    without the offset head, and for models with several input channels (the HCQT shape, 6 x 72, and 3 x 40: convg.hip's two-plane kernel
    writes the a2 planes there)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('split', {}), ('fp32', {'AMTX_X3_NO_SPLIT': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _X3_SPLIT_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['split']), np.load(files['fp32'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 12
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


_FEATS16_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd import tools
from amt_tools_amd.features import HCQT
from amt_tools_amd.models import OnsetsFrames, PendingFeatures16
from amt_tools_amd.synth import synth_clip, synth_state_dict
outs = {}
mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
sd = synth_state_dict(5, dim_in=72, in_channels=6, model_complexity=2)
model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device='cuda:0', precision='bf16')
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model.frontend = torch.nn.Sequential(mod.frontend())
model.change_device()
model.eval()
seen = []
fwd = model.forward
model.forward = lambda feats: (seen.append(type(feats).__name__), fwd(feats))[1]
for name, B, n in (('small', 3, 512 * 45 - 1), ('tiles', 5, 512 * 100 + 17), ('tiny', 2, 4097)):
    audio = torch.from_numpy(np.stack([synth_clip(20 + i, num_samples=n) for i in range(B)]))
    with torch.no_grad():
        out = model.run_on_batch({tools.KEY_AUDIO: audio})
        T = out[tools.KEY_ONSETS].shape[-1]
        lab = model.run_on_batch({tools.KEY_AUDIO: audio, tools.KEY_MULTIPITCH: torch.zeros(B, 88, T), tools.KEY_ONSETS: torch.zeros(B, 88, T)})
    for k in (tools.KEY_ONSETS, tools.KEY_MULTIPITCH):
        outs[f'{name}_roll_{k}'] = out[k].cpu().numpy()
        outs[f'{name}_lab_{k}'] = lab[k].cpu().numpy()
    outs[f'{name}_loss'] = np.asarray(float(lab[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]))
outs['kinds'] = np.asarray(sorted(set(seen)))
np.savez(sys.argv[1], **outs)
'''


def test_hcqt_features_in_the_conv_kernels_staging_format_return_the_same_bits(tmp_path):
    """Round 5, BASELINE config 3 in bf16: inside run_on_batch the HCQT front-end writes its map as (B,T,F,8) bf16 -- the six harmonics of a
    position in one 16-byte slot, the feature tile format of the fused first conv (amtx_cqt_forward16 -> amtx_of_forward_feats16) -- and
    the conv kernel stages a position with one load.  AMTX_CQT_FEATS16=0 keeps the fp32 (B,C,F,T) map the kernel converts itself.  The
    bf16 values are the same roundings either way: rolls, logits (labelled batches) and losses must be IDENTICAL; clips of 45 frames and
    of 101 (seven frame tiles, three column tiles, ragged edges), and of 9 (less than one tile of anything)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('feats16', {}), ('fp32', {'AMTX_CQT_FEATS16': '0'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _FEATS16_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['feats16']), np.load(files['fp32'])
    assert list(a['kinds']) == ['PendingFeatures16'] and list(b['kinds']) == ['Tensor']        # each arm ran the path it names
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 16
    for k in a.files:
        if k != 'kinds':
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a['tiles_roll_onsets'].shape == (5, 88, 101) and 0 < a['tiles_lab_onsets'].std()


def test_cqt_forward16_is_the_fp32_map_rounded_to_bf16_channels_last():
    """amtx_cqt_forward16 against amtx_cqt_forward: out16[b, t, f, h] == bf16(out[b, h, f, t]) bit for bit, slots 6 and 7 zero; dB and linear
    magnitudes, a bin count above one LDS pass (144 > 120), a one-harmonic CQT."""
    from amt_tools_amd.features import CQT, HCQT
    audio = torch.from_numpy(np.stack([synth_clip(40 + i, num_samples=512 * 37 + 5) for i in range(3)])).cuda()
    for mod in (HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12),
                HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, decibels=False),
                CQT(sample_rate=22050, hop_length=512, n_bins=144, bins_per_octave=24)):
        ref = mod.process_batch(audio)                                       # (B, C, F, T) fp32
        got = mod.process_batch16(audio)                                     # (B, T, F, 8) bf16
        C = ref.shape[1]
        assert got.shape == (3, ref.shape[-1], ref.shape[2], 8) and got.dtype == torch.bfloat16
        want = ref.permute(0, 3, 2, 1).to(torch.bfloat16)
        assert torch.equal(got[..., :C].view(torch.int16), want.contiguous().view(torch.int16))
        assert not got[..., C:].view(torch.int16).any()


@pytest.mark.parametrize('F,C,B,T', [(72, 6, 3, 50), (40, 2, 2, 33), (34, 8, 2, 49), (100, 3, 2, 20), (64, 6, 2, 17), (24, 4, 2, 16), (66, 5, 1, 97)])
def test_engine_takes_16_bit_channels_last_features_bit_for_bit(F, C, B, T):
    """amtx_of_forward_feats16 against amtx_of_forward on the same random features: the (B,T,F,8) bf16 map is what the conv kernel rounds the
    fp32 map to itself, so every logit must be IDENTICAL -- widths with a remainder of 8, 2 and 4 columns behind whole 32-column tiles
    (strip tiles: one, two and three 16-frame strips in the last strip tile, one to four valid columns per wave), a multiple of 32 and a
    map narrower than one tile (no strip tiles), 2 .. 8 input channels (slots C .. 7 zero)."""
    from amt_tools_amd.models import OnsetsFrames, PendingFeatures16
    sd = synth_state_dict(3, dim_in=F, in_channels=C, model_complexity=2)
    model = OnsetsFrames(F, tools.PianoProfile(), C, 2, device='cuda:0', precision='bf16')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    model.eval()
    eng = model._get_engine(torch.device('cuda:0'))
    assert eng.takes_feats16()
    rng = np.random.default_rng(F * 100 + C)
    feats = torch.from_numpy(rng.random((B, C, F, T)).astype(np.float32)).cuda()
    feats[0, :, :, : T // 3] = 0.0                                        # silence at the start of a clip
    f16 = torch.zeros((B, T, F, 8), dtype=torch.bfloat16, device='cuda:0')
    f16[..., :C] = feats.permute(0, 3, 2, 1).to(torch.bfloat16)
    with torch.no_grad():
        ref = eng.forward(feats.transpose(-1, -2))
        got = eng.forward(PendingFeatures16(None, f16, None))
    for r, g_, name in zip(ref, got, ('onsets_roll', 'multi_pitch_roll', 'onsets', 'multi_pitch', 'pitch_head')):
        assert torch.equal(r, g_), name
    assert ref[2].std() > 0


def test_feats16_entry_refuses_models_it_is_not_built_for():
    """amtx_of_takes_feats16 is 0 for a one-channel model and for model_complexity 3 in x3, and amtx_of_forward_feats16 then fails loudly instead of
    computing something else; the x3 HCQT model (round 6) answers 2 = the two-plane form."""
    from amt_tools_amd import _lib
    from amt_tools_amd.models import OnsetsFrames, PendingFeatures16
    for F, C, mc, prec in ((72, 6, 3, 'x3'), (229, 1, 2, 'bf16')):
        sd = synth_state_dict(3, dim_in=F, in_channels=C, model_complexity=mc)
        model = OnsetsFrames(F, tools.PianoProfile(), C, mc, device='cuda:0', precision=prec)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.change_device()
        model.eval()
        eng = model._get_engine(torch.device('cuda:0'))
        assert not eng.takes_feats16()
        with pytest.raises(_lib.AmtxError):
            eng.forward(PendingFeatures16(None, torch.zeros((1, 16, F, 8), dtype=torch.bfloat16, device='cuda:0'), None))
    model = OnsetsFrames(72, tools.PianoProfile(), 6, 2, device='cuda:0', precision='x3')
    model.change_device()
    model.eval()
    assert model._get_engine(torch.device('cuda:0')).takes_feats16() == 2


def test_cqt_forward16_split_planes_are_the_split_of_the_fp32_map():
    """amtx_cqt_forward16_split: plane 0 = amtx_cqt_forward16's map (bf16 of the fp32 feature, channels last, slots 6 .. 7 zero), plane 1 =
    bf16(feature - plane 0): the two planes the x3 kernels' own conversion (split_bf16x2) makes -- hi + lo restores the feature to 2^-16."""
    from amt_tools_amd.features import HCQT
    mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    audio = torch.from_numpy(np.stack([synth_clip(i, num_samples=512 * 50 - 3) for i in range(3)])).cuda()
    f32 = mod.process_batch(audio)                                   # (B, 6, 72, T)
    one = mod.process_batch16(audio)                                 # (B, T, 72, 8)
    two = mod.process_batch16(audio, split=True)                     # (2, B, T, 72, 8)
    assert two.shape == (2,) + tuple(one.shape) and torch.equal(two[0], one)
    want = f32.permute(0, 3, 2, 1)                                   # (B, T, 72, 6)
    hi = want.to(torch.bfloat16)
    lo = (want - hi.float()).to(torch.bfloat16)
    assert torch.equal(two[0][..., :6], hi) and torch.equal(two[1][..., :6], lo)
    assert not two[..., 6:].any()
    assert ((two[0].float() + two[1].float())[..., :6] - want).abs().max().item() < 2.0 ** -15


@pytest.mark.parametrize('shape', [(3, 50, 72, 6), (2, 33, 84, 6), (37, 140, 72, 6), (1, 17, 20, 6), (2, 21, 73, 2), (3, 40, 36, 7), (2, 16, 229, 3)])
def test_x3_multichannel_first_conv_on_convx_matches_the_general_kernel(shape, monkeypatch):
    """Round 6: in the x3 precision a model with 2 .. 8 input channels takes its features as the two planes of the split ((2,B,T,F,8),
    amtx_cqt_forward16_split) and runs conv1 + conv2 on convx.hip's layer-specialised kernel (`convx12_kernel<true>`: tap-major first conv by the
    layer1 waves, layer2 by the others) instead of convg.hip's two-plane kernel on fp32 features.  Same fragments, same step and product order in
    both layers' accumulators as the kernel it replaces: the engine's logits must be the bits of the fp32-feature path (AMTX_NO_CONVX12M is read when
    the engine is created).  72 bins = four 18-column tiles, 84 = five narrower ones, 20 = one partial tile, 73 an odd width, 229 thirteen tiles;
    frame counts inside a 16-row tile; 2, 3, 6 and 7 input channels (8 keeps fp32 activations: ofmodel.hip split_acts)."""
    from amt_tools_amd.models import OnsetsFrames, PendingFeatures16
    B, T, F, Cin = shape
    sd = synth_state_dict(9, dim_in=F, in_channels=Cin, model_complexity=2)
    rng = np.random.default_rng(B * 1000 + T)
    feats = torch.from_numpy(rng.random((B, Cin, F, T)).astype(np.float32)).cuda()          # (B, C, F, T)
    want = feats.permute(0, 3, 2, 1)                                                          # (B, T, F, C)
    hi = want.to(torch.bfloat16)
    lo = (want - hi.float()).to(torch.bfloat16)
    planes = torch.zeros((2, B, T, F, 8), dtype=torch.bfloat16, device='cuda')
    planes[0, ..., :Cin], planes[1, ..., :Cin] = hi, lo
    got = {}
    for mode in ('convx', 'convg'):
        if mode == 'convg':
            monkeypatch.setenv('AMTX_NO_CONVX12M', '1')
        else:
            monkeypatch.delenv('AMTX_NO_CONVX12M', raising=False)
        model = OnsetsFrames(F, tools.PianoProfile(), Cin, 2, device='cuda:0', precision='x3')
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.change_device()
        model.eval()
        eng = model._get_engine(torch.device('cuda:0'))
        with torch.no_grad():
            if mode == 'convx':
                assert eng.takes_feats16() == 2
                out = eng.forward(PendingFeatures16(None, planes, None))
            else:
                assert eng.takes_feats16() == 0
                out = eng.forward(feats.transpose(-1, -2))
        got[mode] = [o.clone() for o in out]
        del model
    for a, b in zip(got['convx'], got['convg']):
        assert torch.equal(a, b), (shape, (a - b).abs().max().item())


_STRIP_MC3_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict
outs = {}
sd = synth_state_dict(7, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
model = OnsetsFrames2(229, tools.PianoProfile(), 1, 3, device='cuda:0', precision='bf16')
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model.change_device()
model.eval()
for name, B, T in (('a', 2, 50), ('b', 3, 97), ('c', 1, 16)):
    rng = np.random.default_rng(B * 100 + T)
    feats = torch.from_numpy(rng.random((B, 1, 229, T)).astype(np.float32)).cuda()
    with torch.no_grad():
        lg = model.engine_logits(feats)
    for k, v in lg.items():
        outs[f'{name}_{k}'] = v.cpu().numpy()
np.savez(sys.argv[1], **outs)
'''


def test_strip_tiles_of_the_one_channel_first_conv_return_the_same_bits(tmp_path):
    """OnsetsFrames2 as shipped (model_complexity 3, 229 mel bins): conv2's last column tile holds 4 of its 228 pooled-from columns; those
    run as strip tiles (three 16-frame blocks side by side) in a launch of their own.  AMTX_CONVG_NO_STRIP=1: one launch of 32-column tiles.
    Same arithmetic per output: every logit IDENTICAL; 50, 97 and 16 frames = strip tiles with one, two and three strips."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('strip', {}), ('plain', {'AMTX_CONVG_NO_STRIP': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _STRIP_MC3_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['strip']), np.load(files['plain'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 15
    for k in a.files:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a['a_onsets'].std() > 0
