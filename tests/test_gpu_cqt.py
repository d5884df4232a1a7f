"""GPU parity: HIP CQT / VQT / HCQT front-end (time-domain MFMA formulation, through the C ABI) vs the
frequency-domain fp64 oracle restatement of librosa.vqt.  Tolerance 1e-3 absolute in the scaled [0,1] domain
(the decimation chain dominates; both sides use this project's documented decimator), 2e-4 relative to
the map maximum on linear magnitudes."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import cqt_np as cq                 # noqa: E402
from amt_tools_amd.synth import synth_clip      # noqa: E402

TOL = 1e-3


@pytest.mark.parametrize('lv', ['0.10', '0.9'])
def test_cqt_config1_matches_oracle(lv):
    """TabCNN's front-end (examples/papers/tabcnn.py:80-87): CQT(192 bins, 24 per octave), 8 octaves."""
    from amt_tools_amd.features import CQT
    y = synth_clip(2, num_samples=50000)
    mod = CQT(sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24, librosa_version=lv)
    got = mod.process_audio(y)
    ref = cq.cqt_process_audio(y, sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24, lv=lv)
    assert got.shape == ref.shape == (1, 192, 1 + 50000 // 512) and got.dtype == np.float32
    assert np.abs(got - ref).max() < TOL
    assert got.max() == 1.0 and got.min() >= 0.0


def test_cqt_linear_magnitudes():
    from amt_tools_amd.features import CQT
    y = synth_clip(4, num_samples=30000)
    got = CQT(n_bins=84, decibels=False).process_audio(y)
    ref = cq.cqt_process_audio(y, n_bins=84, decibels=False)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-4 * ref.max()


def test_vqt_default_gamma_matches_oracle():
    from amt_tools_amd.features import VQT
    y = synth_clip(6, num_samples=40000)
    mod = VQT(n_bins=72)
    got = mod.process_audio(y)
    ref = cq.vqt_process_audio(y, n_bins=72)
    assert got.shape == ref.shape and np.abs(got - ref).max() < TOL
    assert mod.get_expected_frames(y) == cq.vqt_expected_frames(len(y), 22050, 512, cq.C1_HZ, 72, 12, mod.gamma)


def test_hcqt_config3_matches_oracle_and_bookkeeping():
    """BASELINE config 3: HCQT(6 harmonics, 72 bins) at 22.05 kHz (84 bins exceed Nyquist)."""
    from amt_tools_amd.features import HCQT
    from amt_tools_amd import _lib
    y = synth_clip(8, num_samples=512 * 60 - 1)
    mod = HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    got = mod.process_audio(y)
    ref = cq.hcqt_process_audio(y, sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)
    assert got.shape == ref.shape == (6, 72, mod.get_expected_frames(y))
    assert np.abs(got - ref).max() < TOL
    assert mod.get_num_channels() == 6 and mod.get_feature_size() == 72
    with pytest.raises(_lib.AmtxError):
        HCQT(n_bins=84).process_audio(y)                      # harmonic 3 would exceed Nyquist


def test_batch_and_full_size_clip_properties():
    from amt_tools_amd.features import HCQT
    mod = HCQT(n_bins=72)
    clips = np.stack([synth_clip(i) for i in range(3)])
    x = torch.from_numpy(clips).cuda()
    full = mod.process_batch(x)
    assert full.shape == (3, 6, 72, 625) and torch.isfinite(full).all()
    solo = mod.process_batch(x[1:2])
    assert torch.equal(full[1], solo[0])                          # clips are independent
    assert torch.all(full.amax(dim=(2, 3)) == 1.0)                # every (clip, harmonic) map is normalised to its own max
    lin = HCQT(n_bins=72, decibels=False)
    a = lin.process_batch(x[:1])
    b = lin.process_batch(4.0 * x[:1])
    assert torch.allclose(b, 4.0 * a, rtol=1e-5, atol=0)          # power-of-two gain: linear in amplitude


def test_config1_cqt_into_tabcnn_end_to_end():
    """BASELINE config 1 on the build's objects: one GuitarSet-shape clip -> CQT(192 bins, 24 per octave) on the GPU (HIP) ->
    TabCNN (stock torch ops) -> tablature (6 strings x T).  Against the oracle's CQT features through the same model on the
    CPU: identical tablature except where the two best classes of a string are within the feature tolerance of each other."""
    torch = pytest.importorskip('torch')
    from amt_tools_amd import tools
    from amt_tools_amd.features import CQT
    from amt_tools_amd.models import TabCNN
    from amt_tools_amd.synth import synth_tabcnn_state_dict
    y = synth_clip(5, num_samples=4 * 22050)
    mod = CQT(sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24)
    feats = mod.process_audio(y)                                                   # (1, 192, T) from the HIP kernels
    ref_feats = cq.cqt_process_audio(y, sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24).astype(np.float32)
    assert feats.shape == ref_feats.shape == (1, 192, mod.get_expected_frames(y))
    profile = tools.GuitarProfile(num_frets=19)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_tabcnn_state_dict(3, dim_in=192).items()}
    outs, logits = [], []
    for f, dev in ((feats, 'cuda:0'), (ref_feats, 'cpu')):
        model = TabCNN(192, profile, 1, 1, device=dev)
        model.load_state_dict(sd)
        model.change_device()
        model.eval()
        with torch.no_grad():
            batch = {tools.KEY_FEATS: torch.from_numpy(f[None])}
            raw = model(model.pre_proc(dict(batch))[tools.KEY_FEATS])[tools.KEY_TABLATURE]
            outs.append(model.run_on_batch(batch)[tools.KEY_TABLATURE].cpu().numpy())
            logits.append(raw.cpu().numpy())
    assert outs[0].shape == (1, 6, feats.shape[-1]) and outs[0].min() >= -1 and outs[0].max() <= 19
    assert np.abs(logits[0] - logits[1]).max() < 5e-2
    top2 = np.sort(logits[1].reshape(1, -1, 6, 21), axis=-1)[..., -2:]
    decided = np.swapaxes(top2[..., 1] - top2[..., 0], -1, -2) > 0.1               # (1, 6, T): argmax not a near-tie
    assert np.all((outs[0] == outs[1]) | ~decided) and decided.mean() > 0.5


_BASIS_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd.features import CQT, HCQT, VQT
from amt_tools_amd.synth import synth_clip
outs = {}
y = np.stack([synth_clip(i, num_samples=70001) for i in range(3)])
y[1] *= 1e-3
for name, mod in (('cqt1', CQT(sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24)),
                  ('cqt1_09', CQT(sample_rate=22050, hop_length=512, n_bins=192, bins_per_octave=24, librosa_version='0.9')),
                  ('hcqt3', HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12)),
                  ('vqt', VQT(sample_rate=22050, hop_length=256, n_bins=60, bins_per_octave=12, gamma=5.0)),
                  ('hcqt_lin', HCQT(sample_rate=22050, hop_length=512, n_bins=72, bins_per_octave=12, decibels=False))):
    outs[name] = mod.process_batch(torch.from_numpy(y).cuda()).cpu().numpy()
    outs[name + '_one'] = mod.process_audio(y[2][:33333])
    if name in ('cqt1', 'hcqt3'):                   # short clips: every level a single (edge) tile, the deepest a few dozen samples; five clips
        ys = np.stack([synth_clip(9 + i, num_samples=9001) for i in range(5)])
        outs[name + '_short'] = mod.process_batch(torch.from_numpy(ys).cuda()).cpu().numpy()
    if name in ('cqt1', 'hcqt3', 'cqt1_09'):        # rows on 16-byte boundaries (the decimator's vector staging), several tiles per clip and level
        ya = np.stack([synth_clip(5 + i, num_samples=81920) for i in range(2)])
        outs[name + '_al'] = mod.process_batch(torch.from_numpy(ya).cuda()).cpu().numpy()
np.savez(sys.argv[1], **outs)
'''


def test_windowed_basis_kernel_returns_the_bits_of_the_gemm_path(tmp_path):
    """Round 5: the per-level basis products run on cqt_basis_kernel (a tile's signal staged and split into its two 16-bit planes ONCE in
    LDS, rows = overlapping 16-byte windows of it, the basis in registers); AMTX_CQT_GEMM_BASIS=1 keeps the generic fp32-A two-plane GEMM.
    Same planes, same product order, same epilogue: CQT (8 levels, hops 512 .. 4, 48 columns per level), HCQT (banks of several
    harmonics per level), a VQT, both librosa padding conventions, dB and linear output must be IDENTICAL."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('windowed', {}), ('gemm', {'AMTX_CQT_GEMM_BASIS': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _BASIS_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['windowed']), np.load(files['gemm'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 15
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].size > 0
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_persistent_decimator_returns_the_bits_of_round_4s(tmp_path):
    """Round 5: the half-band decimations of the pyramid run on cqt_decimate2_kernel (persistent blocks, 16-byte staging of the tiles
    inside a clip, two output blocks per matrix column sharing their fragments, swizzled planes); AMTX_CQT_DECIM_V1=1 keeps round 4's
    kernel.  Every output accumulates the same products in the same order: whole CQT / HCQT / VQT maps must be IDENTICAL -- clips whose
    rows are 16-byte aligned (vector staging from the caller's audio) and not (70001 samples: scalar staging there, vector staging from
    the pyramid levels), both padding conventions."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('v2', {}), ('v1', {'AMTX_CQT_DECIM_V1': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _BASIS_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['v2']), np.load(files['v1'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 15
    for k in a.files:
        assert a[k].shape == b[k].shape and a[k].size > 0
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_one_clip_batch_with_a_degenerate_batch_stride():
    """`torch.from_numpy(y[None])` has batch stride 0 (NumPy's new axis) and torch calls it contiguous: a one-clip batch must not hand that
    stride to the C ABI (found by the round-6 full-size check; amtx_cqt_forward / amtx_spec_power rejected it as audio_stride < num_samples)."""
    from amt_tools_amd.features import HCQT, MelSpec
    y = synth_clip(3, num_samples=512 * 40 - 1)
    x = torch.from_numpy(y[None]).cuda()
    ref = torch.from_numpy(y.copy()).cuda().reshape(1, -1)
    for mod in (HCQT(n_bins=72), MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)):
        assert torch.equal(mod.process_batch(x), mod.process_batch(ref))
