"""Training path on the GPU: the HIP BiLSTM autograd function against torch's own LSTM (forward values and every gradient),
and one whole training step of the model against the reference's golden losses / gradients (tests/golden/of1_train.npz).
Tolerances: the recurrent mat-vecs use split-bf16 (fp32-class) products -> 2e-4 relative to the largest element."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from conftest import load_golden                 # noqa: E402
from amt_tools_amd import tools                  # noqa: E402


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


@pytest.mark.parametrize('H', [128, 256])
@pytest.mark.parametrize('B,T,I', [(1, 1, 16), (3, 20, 64), (8, 37, 176), (5, 64, 512)])
def test_bilstm_autograd_matches_torch_lstm(B, T, I, H):
    """hidden 128: register-stationary kernels; hidden 256 (model_complexity 3): the streaming forward / backward kernels."""
    from amt_tools_amd.autograd import bilstm
    torch.manual_seed(B * 1000 + T)
    ref = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).double()
    x = torch.randn(B, T, I, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(B, T, 2 * H, dtype=torch.float64)
    y_ref = ref(x)[0]
    y_ref.backward(gy)
    mine = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).cuda()
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    xc = x.detach().float().cuda().requires_grad_(True)
    y = bilstm(xc, mine)
    y.backward(gy.float().cuda())
    assert _rel(y.detach().cpu().double(), y_ref.detach()) < 2e-5
    assert _rel(xc.grad.cpu().double(), x.grad) < 2e-4
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad.cpu().double(), q.grad) < 2e-4, n


def test_training_step_on_gpu_matches_reference_golden():
    """The reference's training-mode losses and gradients (BatchNorm batch statistics, dropout off) with the model on the GPU:
    ATen convolutions + the HIP BiLSTM forward/backward."""
    from amt_tools_amd.models import OnsetsFrames
    from amt_tools_amd.synth import synth_state_dict
    g = load_golden('of1_train.npz')
    model = OnsetsFrames(229, tools.PianoProfile(), 1, int(g['model_complexity']), device='cuda:0')
    sd = synth_state_dict(int(g['seed']), dim_in=229, in_channels=1, model_complexity=int(g['model_complexity']))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.change_device()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    batch = {tools.KEY_FEATS: torch.from_numpy(g['feats']), tools.KEY_MULTIPITCH: torch.from_numpy(g['multi_pitch']),
             tools.KEY_ONSETS: torch.from_numpy(g['onsets'])}
    out = model.run_on_batch(batch)
    loss = out[tools.KEY_LOSS]
    assert abs(loss[tools.KEY_LOSS_PITCH].item() - float(g['loss_pitch'])) < 2e-3
    assert abs(loss[tools.KEY_LOSS_ONSETS].item() - float(g['loss_onsets'])) < 2e-3
    loss[tools.KEY_LOSS_TOTAL].backward()
    named = dict(model.named_parameters())
    for i, k in enumerate(g['grad_keys']):
        ref = g[f'grad_{i}']
        got = named[str(k)].grad.cpu().numpy()
        assert np.abs(got - ref).max() / max(1e-6, np.abs(ref).max()) < 5e-3, k
    # the stock path gives the same numbers (switch kept for A/B)
    for mod in model.modules():
        if hasattr(mod, 'use_hip_autograd'):
            mod.use_hip_autograd = False
        if hasattr(mod, 'use_hip_bn'):
            mod.use_hip_bn = False
    model.zero_grad()
    out2 = model.run_on_batch(batch)
    assert abs(out2[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].item() - loss[tools.KEY_LOSS_TOTAL].item()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,K,weighted', [(8, 625, 88, False), (3, 70, 88, True), (1, 1, 4, False), (2, 33, 120, True)])
def test_bce_logits_loss_matches_torch(B, T, K, weighted):
    """amtx_bce_logits_loss = LogisticBank.get_loss (common.py:541-584) and its gradient, against torch in float64."""
    import torch.nn.functional as F
    from amt_tools_amd.autograd import bce_logits_loss
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = (torch.randn(B, T, K, generator=g) * 4).cuda().requires_grad_(True)
    y = (torch.rand(B, K, T, generator=g) < 0.1).float().cuda()
    w = (torch.rand(K, generator=g) + 0.5).cuda() if weighted else None
    loss = bce_logits_loss(x, y, w)
    (loss * 3.0).backward()
    x64 = x.detach().double().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(x64.transpose(-2, -1), y.double(), weight=None if w is None else w.double().unsqueeze(-1),
                                             reduction='none').mean(dim=-1).sum(dim=-1).mean()
    (ref * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-7          # tolerance: fp32 terms, double reduction
    assert (x.grad.double() - x64.grad).abs().max().item() <= 1e-6 * x64.grad.abs().max().item() + 1e-9
    again = bce_logits_loss(x.detach(), y, w)
    assert again.item() == loss.item()                                             # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,T,F,pool', [(2, 32, 9, 229, False), (2, 32, 9, 229, True), (3, 64, 5, 114, True), (1, 48, 7, 18, True),
                                          (1, 96, 3, 2, True), (2, 4, 3, 5, False), (8, 32, 64, 229, True)])
def test_bn_relu_pool_training_matches_torch(B, C, T, F, pool):
    """amtx_bn_relu_pool_train_fwd / _bwd = nn.BatchNorm2d (training) -> ReLU -> [MaxPool2d((1,2))] against torch in float64:
    output, input gradient, gamma / beta gradients, running statistics, num_batches_tracked."""
    from amt_tools_amd.autograd import bn_relu_pool, bn_relu_pool_supported
    gen = torch.Generator().manual_seed(B * 100 + C + T + F)
    x = (torch.randn(B, C, T, F, generator=gen) * 1.5 + 0.3)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=gen) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=gen) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=gen) + 0.5)
    import copy
    ref_bn = copy.deepcopy(bn).double()
    x64 = x.double().requires_grad_(True)
    y_ref = torch.relu(ref_bn(x64))
    if pool:
        y_ref = torch.nn.functional.max_pool2d(y_ref, (1, 2))
    gy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(gy.double())

    bn = bn.cuda()
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert bn_relu_pool_supported(xc, bn)
    y = bn_relu_pool(xc, bn, pool)
    assert y.shape == y_ref.shape
    y.backward(gy.cuda())
    assert _rel(y.detach().cpu().double(), y_ref.detach()) < 2e-6
    assert _rel(xc.grad.cpu().double(), x64.grad) < 2e-5
    assert _rel(bn.weight.grad.cpu().double(), ref_bn.weight.grad) < 2e-5
    assert _rel(bn.bias.grad.cpu().double(), ref_bn.bias.grad) < 2e-5
    assert _rel(bn.running_mean.cpu().double(), ref_bn.running_mean) < 1e-6
    assert _rel(bn.running_var.cpu().double(), ref_bn.running_var) < 1e-6
    assert int(bn.num_batches_tracked) == 1
    # deterministic (no atomics)
    xc2 = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y2 = bn_relu_pool(xc2, bn, pool)
    y2.backward(gy.cuda())
    assert torch.equal(xc2.grad, xc.grad)
