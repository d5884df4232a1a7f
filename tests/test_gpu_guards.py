"""Guard bands around every kernel workspace.  Several kernels deliberately issue masked stores to scratch lines and
loads from clamped addresses (conv.hip, convf.hip, gemm.hip); a store that leaves its workspace corrupts a neighbouring tensor silently.
With _lib.GUARD_BYTES set, the engine, CQT and training workspaces are allocated with 4 KiB of pattern on each side; after forward (and
backward) passes over the ragged shapes the suite uses elsewhere the pattern must be untouched."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from amt_tools_amd import _lib, tools                          # noqa: E402
from amt_tools_amd.synth import synth_state_dict, synth_clip   # noqa: E402


@pytest.fixture
def guards(monkeypatch):
    monkeypatch.setattr(_lib, 'GUARD_BYTES', 4096)
    yield


@pytest.mark.parametrize('cls,mc,ch,dim_in,precision', [('OnsetsFrames', 2, 1, 229, 'bf16'), ('OnsetsFrames', 2, 1, 229, 'x3'), ('OnsetsFrames', 2, 1, 229, 'f16'),
                                                        ('OnsetsFrames2', 3, 1, 229, 'bf16'), ('OnsetsFrames2', 3, 1, 229, 'f16'), ('OnsetsFrames', 2, 6, 72, 'bf16'), ('OnsetsFrames', 2, 6, 72, 'f16'), ('OnsetsFrames', 2, 1, 54, 'bf16'),
                                                        ('OnsetsFrames', 2, 1, 8, 'x3'),
                                                        # convg.hip's weight-chunk modes: two DMA buffers (4, 5 in one plane), one chunk per block (two planes)
                                                        ('OnsetsFrames', 4, 1, 229, 'bf16'), ('OnsetsFrames', 5, 1, 229, 'bf16'), ('OnsetsFrames2', 3, 1, 229, 'x3'),
                                                        ('OnsetsFrames', 4, 3, 72, 'x3')])
def test_engine_workspace_guard_bands_survive_ragged_forwards(guards, cls, mc, ch, dim_in, precision):
    import amt_tools_amd.models as M
    sd = synth_state_dict(7, dim_in=dim_in, in_channels=ch, model_complexity=mc, offsets=cls == 'OnsetsFrames2')
    rng = np.random.default_rng(dim_in)
    for B, T in ((1, 1), (3, 17), (17, 9), (2, 40), (1, 33), (130, 47), (44, 140)):
        model = getattr(M, cls)(dim_in, tools.PianoProfile(), ch, mc, device='cuda:0', precision=precision)     # fresh engine: buffer == need
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.change_device()
        model.eval()
        feats = torch.from_numpy(rng.random((B, ch, dim_in, T)).astype(np.float32)).cuda()
        with torch.no_grad():
            model.engine_logits(feats)
            model.run_on_batch({tools.KEY_FEATS: feats})
        torch.cuda.synchronize()
        eng = model._get_engine(feats.device)
        assert eng.workspace._base is not None and eng.workspace.data_ptr() % 256 == 0
        assert _lib.guards_intact(eng.workspace), (cls, precision, B, T)


def test_power_path_and_cqt_workspace_guard_bands(guards):
    from amt_tools_amd.features import MelSpec, HCQT
    import amt_tools_amd.models as M
    mod = MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048)
    sd = synth_state_dict(3, dim_in=229, in_channels=1, model_complexity=2)
    for B, n in ((3, 512 * 20), (140, 512 * 60), (2, 5000)):
        model = M.OnsetsFrames(229, tools.PianoProfile(), 1, 2, device='cuda:0', precision='bf16')
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.frontend = torch.nn.Sequential(mod.frontend())
        model.change_device()
        model.eval()
        audio = torch.from_numpy(np.stack([synth_clip(i, num_samples=n) for i in range(B)])).cuda()
        with torch.no_grad():
            model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        assert _lib.guards_intact(model._get_engine(audio.device).workspace), (B, n)
    cq = HCQT(sample_rate=22050, hop_length=512, fmin=32.7, harmonics=[0.5, 1, 2, 3, 4, 5], n_bins=72, bins_per_octave=12)
    for B, n in ((1, 22050), (3, 40000), (2, 512 * 33 + 7)):
        cq.__dict__.pop('_workspace', None)
        audio = torch.from_numpy(np.stack([synth_clip(i, num_samples=n) for i in range(B)])).cuda()
        cq.process_batch(audio)
        torch.cuda.synchronize()
        ws = cq.__dict__.get('_workspace')
        assert ws is not None and ws._base is not None and _lib.guards_intact(ws), (B, n)


def test_training_scratch_guard_bands(guards):
    """One training step (fwd + bwd) of both model families on the HIP autograd kernels with the per-stream scratch buffer guarded."""
    import amt_tools_amd.models as M
    from amt_tools_amd import autograd
    from amt_tools_amd.synth import synth_labels
    autograd._WS.clear()
    for cls, mc in (('OnsetsFrames', 2), ('OnsetsFrames2', 3)):
        torch.manual_seed(0)
        model = getattr(M, cls)(229, tools.PianoProfile(), 1, mc, device='cuda:0')
        model.change_device()
        model.train()
        B, T = 2, 50
        feats = torch.rand(B, 1, 229, T, device='cuda')
        lab = torch.from_numpy((np.random.default_rng(1).random((B, 88, T)) < 0.05).astype(np.float32)).cuda()
        batch = {tools.KEY_FEATS: feats, tools.KEY_MULTIPITCH: lab, tools.KEY_ONSETS: lab.clone()}
        if cls == 'OnsetsFrames2':
            batch[tools.KEY_OFFSETS] = lab.clone()
        model.run_on_batch(batch)[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
        torch.cuda.synchronize()
    assert autograd._WS, 'the HIP autograd path did not run'
    for ws in autograd._WS.values():
        assert ws._base is not None and _lib.guards_intact(ws)
    autograd._WS.clear()
