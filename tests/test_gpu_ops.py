"""GPU parity of the individual HIP kernels (called through the C ABI) against plain fp32 torch-CPU
references of the same op.  Tolerances: 'x3' (split-bf16) is held to fp32-class error; 'bf16' to the
rounding of bf16 operands (2^-9 relative per operand) accumulated in fp32."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F   # noqa: E402

pytestmark = pytest.mark.gpu

from amt_tools_amd import _lib   # noqa: E402

BF16, F32 = 0, 1


def _stream():
    return _lib.current_stream()


def _to_act(x, planes):
    """fp32 tensor -> device activation tensor of the mode's storage type."""
    return x.cuda().to(torch.bfloat16 if planes == 1 else torch.float32).contiguous()


def _tol(planes, scale):
    return (2e-5 if planes == 2 else 1.5e-2) * scale


@pytest.mark.parametrize('planes', [1, 2])
@pytest.mark.parametrize('m,n,k', [(300, 512, 3648), (1000, 1024, 512), (257, 88, 256), (64, 1024, 176), (5, 88, 512),
                                   (2500, 88, 256), (1300, 88, 3648), (4099, 88, 512), (1024, 4, 128)])      # (the skinny-N kernel: N <= 128, M >= 1024)
def test_linear(planes, m, n, k):
    L = _lib.lib()
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bias = torch.randn(n, generator=g)
    packed = np.zeros(L.amtx_linear_packed_elems(n, k, planes), dtype=np.uint16)
    wn = w.numpy()
    _lib.check(L.amtx_linear_pack(_lib.ptr(wn), n, k, planes, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    a_d = _to_act(a, planes)
    ref = F.linear(a_d.float().cpu().double(), w.double(), bias.double()).float()
    for c_type in ([F32, BF16] if planes == 1 else [F32]):
        ldc = n + 8
        c = torch.zeros(m, ldc, dtype=torch.float32 if c_type == F32 else torch.bfloat16, device='cuda')
        bias_d = bias.cuda()
        _lib.check(L.amtx_linear_fwd(_lib.ptr(a_d), k, BF16 if planes == 1 else F32, _lib.ptr(wp), planes, _lib.ptr(bias_d),
                                     _lib.ptr(c), ldc, c_type, m, n, k, _stream()), 'amtx_linear_fwd')
        got = c[:, :n].float().cpu()
        tol = _tol(planes, ref.abs().max().item()) + (2e-2 if c_type == BF16 else 0)
        assert (got - ref).abs().max().item() < tol
        assert torch.all(c[:, n:] == 0)                       # padding columns untouched
    if planes == 1:   # fp32 A with bf16 arithmetic (the adjoin projection reads fp32 logits)
        a32 = a.cuda()
        c = torch.zeros(m, n, device='cuda')
        _lib.check(L.amtx_linear_fwd(_lib.ptr(a32), k, F32, _lib.ptr(wp), 1, None, _lib.ptr(c), n, F32, m, n, k, _stream()))
        ref2 = F.linear(a.bfloat16().double(), w.bfloat16().double()).float()
        assert (c.cpu() - ref2).abs().max().item() < 1e-3 * ref2.abs().max().item()


SPLIT = 2


@pytest.mark.parametrize('m,n,k', [(300, 512, 3648), (1000, 1024, 512), (777, 1024, 192), (256, 256, 64), (257, 88, 256), (64, 1024, 192), (5, 88, 512),
                                   (2500, 88, 256), (1300, 88, 3648), (4099, 88, 512), (70000, 88, 64)])     # (the skinny-N two-plane kernel: N <= 128, M >= 1024)
def test_linear_split_planes(m, n, k):
    """The x3 precision's own activation format (round 5): A as two 16-bit planes (amtx_split_planes), two-plane weights.  Whole 256-column
    tiles with M >= 256 run on the direct-to-LDS two-plane kernel (gemm_split_kernel), N <= 128 with M >= 1024 on gemm_skinny_split_kernel, the
    rest on the generic kernel's split-A loader.
    Checked against (1) a float64 product of the fp32 values -- fp32-class tolerance -- and (2) the fp32-A two-plane path
    (amtx_linear_fwd, planes = 2) on the same values: the SAME BITS (same planes, same product order hi.hi, hi.lo, lo.hi per 32-deep step);
    (3) the split C epilogue returns hi + lo == the fp32 C rounded the way split_bf16x2 does."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bias = torch.randn(n, generator=g)
    packed = np.zeros(L.amtx_linear_packed_elems(n, k, 2), dtype=np.uint16)
    _lib.check(L.amtx_linear_pack(_lib.ptr(w.numpy()), n, k, 2, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    a_d, bias_d = a.cuda(), bias.cuda()
    lda = k + 8
    planes = torch.full((2, m, lda), 0x7fc0, dtype=torch.int16, device='cuda')          # NaN canaries in the pad columns: never read
    _lib.check(L.amtx_split_planes(_lib.ptr(a_d), k, k, _lib.ptr(planes), lda, m * lda, m, _stream()), 'amtx_split_planes')
    hi = (planes[0, :, :k].to(torch.int32) << 16).view(torch.float32)
    lo = (planes[1, :, :k].to(torch.int32) << 16).view(torch.float32)
    assert (hi + lo - a_d).abs().max().item() <= 2.0 ** -16 * a_d.abs().max().item()
    ref = F.linear(a.double(), w.double(), bias.double()).float()
    c32 = torch.zeros(m, n, device='cuda')
    _lib.check(L.amtx_linear_fwd(_lib.ptr(a_d), k, F32, _lib.ptr(wp), 2, _lib.ptr(bias_d), _lib.ptr(c32), n, F32, m, n, k, _stream()))
    ldc = n + 8
    c = torch.zeros(m, ldc, device='cuda')
    _lib.check(L.amtx_linear_fwd_split(_lib.ptr(planes), lda, m * lda, _lib.ptr(wp), _lib.ptr(bias_d), _lib.ptr(c), ldc, F32, 0, m, n, k, _stream()),
               'amtx_linear_fwd_split')
    assert (c[:, :n].cpu() - ref).abs().max().item() < _tol(2, ref.abs().max().item())
    assert torch.equal(c[:, :n], c32), (c[:, :n] - c32).abs().max().item()
    assert torch.all(c[:, n:] == 0)
    if n % 4 == 0:                                                                     # both kernels have the two-plane C epilogue
        cs = torch.zeros(2, m, ldc, dtype=torch.int16, device='cuda')
        _lib.check(L.amtx_linear_fwd_split(_lib.ptr(planes), lda, m * lda, _lib.ptr(wp), _lib.ptr(bias_d), _lib.ptr(cs), ldc, SPLIT, m * ldc, m, n, k,
                                           _stream()), 'amtx_linear_fwd_split (split C)')
        chi = (cs[0, :, :n].to(torch.int32) << 16).view(torch.float32)
        clo = (cs[1, :, :n].to(torch.int32) << 16).view(torch.float32)
        assert torch.equal(chi, c32.bfloat16().float())                               # hi plane = round-to-nearest-even bf16 of the fp32 result
        assert torch.equal(clo, (c32 - chi).bfloat16().float())
        assert torch.all(cs[:, :, n:] == 0)


def _conv_ref(x_btfc, w, scale, shift):
    """x (B,T,F,C) channels-last -> conv3x3 pad1 (no bias) * scale + shift -> ReLU -> MaxPool(1,2) -> (B,T,F/2,Co)."""
    x = x_btfc.permute(0, 3, 1, 2).double()
    y = F.conv2d(x, (w * scale[:, None, None, None]).double(), padding=1) + shift.double()[None, :, None, None]
    y = F.max_pool2d(F.relu(y), (1, 2))
    return y.permute(0, 2, 3, 1).float()


@pytest.mark.parametrize('planes', [1, 2])
@pytest.mark.parametrize('b,t,f,cout', [(2, 40, 229, 32), (1, 17, 114, 64), (3, 5, 36, 32), (1, 33, 18, 64), (1, 1, 2, 32)])
def test_conv3x3_bn_relu_pool(planes, b, t, f, cout):
    L = _lib.lib()
    g = torch.Generator().manual_seed(b * 1000 + t * 10 + f + cout)
    x = torch.rand(b, t, f, 32, generator=g)
    w = torch.randn(cout, 32, 3, 3, generator=g) / 17.0
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    packed = np.zeros(L.amtx_conv3x3_packed_elems(cout, planes), dtype=np.uint16)
    _lib.check(L.amtx_conv3x3_pack(_lib.ptr(w.numpy()), _lib.ptr(scale.numpy()), cout, planes, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x_d = _to_act(x, planes)
    out = torch.full((b, t, f // 2, cout), -7.0, dtype=x_d.dtype, device='cuda')
    shift_d = shift.cuda()
    _lib.check(L.amtx_conv3x3_fwd(_lib.ptr(x_d), BF16 if planes == 1 else F32, _lib.ptr(wp), planes, _lib.ptr(shift_d),
                                  _lib.ptr(out), b, t, f, cout, _stream()), 'amtx_conv3x3_fwd')
    ref = _conv_ref(x_d.float().cpu(), w, scale, shift)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < _tol(planes, ref.abs().max().item()) + (1e-2 * ref.abs().max().item() if planes == 1 else 0), err


@pytest.mark.parametrize('planes', [1, 2])
@pytest.mark.parametrize('b,t,f,cin,cout', [(2, 40, 229, 48, 48), (1, 17, 114, 48, 96), (3, 5, 36, 48, 48), (1, 33, 18, 48, 96),
                                            (1, 1, 2, 48, 48), (2, 16, 33, 48, 96),
                                            (2, 40, 229, 64, 64), (1, 17, 114, 64, 128), (3, 5, 36, 64, 64), (1, 33, 18, 64, 128), (1, 1, 2, 64, 64),
                                            (2, 40, 229, 80, 80), (1, 17, 114, 80, 160), (3, 5, 36, 80, 80), (1, 1, 2, 80, 160)])
def test_conv3x3_general_channels(planes, b, t, f, cin, cout):
    """convg.hip (weights staged in LDS): the 48 -> 48 and 48 -> 96 layers of model_complexity 3, 64 -> 64 and 64 -> 128 of 4, 80 -> 80 and 80 -> 160 of 5."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(b * 1000 + t * 10 + f + cout)
    x = torch.rand(b, t, f, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9.0 * cin) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    n = L.amtx_conv3x3g_packed_elems(cin, cout, planes)
    assert n > 0
    packed = np.zeros(n, dtype=np.uint16)
    _lib.check(L.amtx_conv3x3g_pack(_lib.ptr(w.numpy()), _lib.ptr(scale.numpy()), cin, cout, planes, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x_d = _to_act(x, planes)
    out = torch.full((b, t, f // 2, cout), -7.0, dtype=x_d.dtype, device='cuda')
    shift_d = shift.cuda()
    _lib.check(L.amtx_conv3x3g_fwd(_lib.ptr(x_d), BF16 if planes == 1 else F32, _lib.ptr(wp), planes, _lib.ptr(shift_d),
                                   _lib.ptr(out), b, t, f, cin, cout, _stream()), 'amtx_conv3x3g_fwd')
    ref = _conv_ref(x_d.float().cpu(), w, scale, shift)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < _tol(planes, ref.abs().max().item()) + (1e-2 * ref.abs().max().item() if planes == 1 else 0), err


def test_conv3x3_general_rejects_unbuilt_channel_counts():
    L = _lib.lib()
    assert L.amtx_conv3x3g_packed_elems(40, 48, 1) == 0 and L.amtx_conv3x3g_packed_elems(48, 80, 1) == 0 and L.amtx_conv3x3g_packed_elems(64, 80, 1) == 0
    dummy = np.zeros(16, dtype=np.float32)
    assert L.amtx_conv3x3g_pack(_lib.ptr(dummy), None, 40, 48, 1, _lib.ptr(dummy)) < 0
    assert b'not built' in L.amtx_last_error()


@pytest.mark.parametrize('cin,f', [(1, 229), (6, 72)])
@pytest.mark.parametrize('layout', ['bcft', 'bctf'])
def test_conv1(cin, f, layout):
    L = _lib.lib()
    b, t, cout = 2, 19, 32
    g = torch.Generator().manual_seed(cin + f)
    x = torch.rand(b, cin, f, t, generator=g)                     # reference layout (B,C,F,T)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 3.0
    shift = torch.randn(cout, generator=g) * 0.1
    xd = x.cuda() if layout == 'bcft' else x.transpose(-1, -2).contiguous().cuda().transpose(-1, -2)
    sb, sc, sf, st = xd.stride()
    out = torch.empty(b, t, f, cout, device='cuda')
    w_d, shift_d = w.cuda(), shift.cuda()          # keep device copies alive across the launch
    _lib.check(L.amtx_conv1_fwd(_lib.ptr(xd), sb, sc, st, sf, _lib.ptr(w_d), _lib.ptr(shift_d), _lib.ptr(out), F32,
                                b, t, f, cin, cout, _stream()), 'amtx_conv1_fwd')
    ref = F.relu(F.conv2d(x.transpose(-1, -2).double(), w.double(), shift.double(), padding=1)).permute(0, 2, 3, 1).float()
    assert (out.cpu() - ref).abs().max().item() < 2e-5


def _lstm_ref(xproj, whh_f, whh_b, H=128):
    """xproj (B,T,2,4H) -> (B,T,2H), explicit loop in fp64."""
    B, T = xproj.shape[:2]
    out = torch.zeros(B, T, 2 * H, dtype=torch.float64)
    for d, whh in enumerate((whh_f.double(), whh_b.double())):
        h = torch.zeros(B, H, dtype=torch.float64)
        c = torch.zeros(B, H, dtype=torch.float64)
        for s in range(T):
            t = s if d == 0 else T - 1 - s
            gates = xproj[:, t, d].double() + h @ whh.T
            i, f, g, o = gates.chunk(4, dim=-1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[:, t, H * d:H * (d + 1)] = h
    return out.float()


@pytest.mark.parametrize('planes', [1, 2])
@pytest.mark.parametrize('b,t', [(1, 1), (3, 50), (17, 33), (40, 120)])
@pytest.mark.parametrize('H', [256, 384, 512])
def test_bilstm_hidden_256_384(planes, b, t, H):
    """lstm.hip bilstm_stream_kernel (W_hh streamed from L2): the recurrences of model_complexity 3, 4 and 5 (hidden 256 / 384 / 512 per direction)."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(b * 100 + t)
    xproj = torch.randn(b, t, 2, 4 * H, generator=g)
    whh_f = (torch.rand(4 * H, H, generator=g) - 0.5) * 0.2
    whh_b = (torch.rand(4 * H, H, generator=g) - 0.5) * 0.2
    packed = np.zeros(L.amtx_bilstm_h_packed_elems(H, planes), dtype=np.uint16)
    _lib.check(L.amtx_bilstm_h_pack(_lib.ptr(whh_f.numpy()), _lib.ptr(whh_b.numpy()), H, planes, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x_d = _to_act(xproj, planes)
    out = torch.full((b, t, 2 * H), 9.0, dtype=x_d.dtype, device='cuda')
    _lib.check(L.amtx_bilstm_h_fwd(_lib.ptr(x_d), _lib.ptr(wp), H, planes, BF16 if planes == 1 else F32, _lib.ptr(out), b, t, _stream()),
               'amtx_bilstm_h_fwd')
    ref = _lstm_ref(x_d.float().cpu(), whh_f, whh_b, H)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < (3e-5 if planes == 2 else 3e-2), err


def test_bilstm_h_entry_runs_hidden_128_like_the_fixed_entry():
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    b, t = 5, 20
    xproj = torch.randn(b, t, 2, 512, generator=g)
    whh_f = (torch.rand(512, 128, generator=g) - 0.5) * 0.3
    whh_b = (torch.rand(512, 128, generator=g) - 0.5) * 0.3
    packed = np.zeros(L.amtx_bilstm_h_packed_elems(128, 2), dtype=np.uint16)
    _lib.check(L.amtx_bilstm_h_pack(_lib.ptr(whh_f.numpy()), _lib.ptr(whh_b.numpy()), 128, 2, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x_d = xproj.cuda()
    out_a = torch.zeros(b, t, 256, device='cuda')
    out_b = torch.zeros(b, t, 256, device='cuda')
    _lib.check(L.amtx_bilstm_h_fwd(_lib.ptr(x_d), _lib.ptr(wp), 128, 2, F32, _lib.ptr(out_a), b, t, _stream()))
    _lib.check(L.amtx_bilstm_fwd(_lib.ptr(x_d), _lib.ptr(wp), 2, F32, _lib.ptr(out_b), b, t, _stream()))
    assert torch.equal(out_a, out_b)
    assert L.amtx_bilstm_h_packed_elems(192, 1) == 0


@pytest.mark.parametrize('planes', [1, 2])
# b <= 2048 runs the four-clips-per-block kernel, larger batches the sixteen-clip one (lstm.hip: use_four_clip_blocks)
# and from 513 clips on (four-clip blocks would outnumber the 256 CUs) the eight-clips-per-block mapping bilstm4_kernel<.., NC = 2>: both
# sides of every batch-size threshold of the dispatch are here (512 | 513, 2048 | 2049), with ragged last blocks (521 = 65 x 8 + 1,
# 2047) and the clip_ok[1] = false tail (513, 521: the last block holds one clip)
@pytest.mark.parametrize('b,t', [(1, 1), (3, 50), (17, 33), (32, 200), (512, 7), (513, 7), (521, 40), (1024, 7), (1030, 40), (2047, 5),
                                 (2048, 7), (2049, 7), (2063, 3)])
def test_bilstm(planes, b, t):
    L = _lib.lib()
    g = torch.Generator().manual_seed(b * 100 + t)
    xproj = torch.randn(b, t, 2, 512, generator=g)
    whh_f = (torch.rand(512, 128, generator=g) - 0.5) * 0.3
    whh_b = (torch.rand(512, 128, generator=g) - 0.5) * 0.3
    packed = np.zeros(L.amtx_bilstm_packed_elems(planes), dtype=np.uint16)
    _lib.check(L.amtx_bilstm_pack(_lib.ptr(whh_f.numpy()), _lib.ptr(whh_b.numpy()), planes, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    x_d = _to_act(xproj, planes)
    out = torch.full((b, t, 256), 9.0, dtype=x_d.dtype, device='cuda')
    _lib.check(L.amtx_bilstm_fwd(_lib.ptr(x_d), _lib.ptr(wp), planes, BF16 if planes == 1 else F32, _lib.ptr(out), b, t, _stream()),
               'amtx_bilstm_fwd')
    ref = _lstm_ref(x_d.float().cpu(), whh_f, whh_b)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < (3e-5 if planes == 2 else 3e-2), err


def test_pianoroll_threshold_and_probabilities():
    L = _lib.lib()
    b, t, keys, ld, col0 = 3, 70, 88, 176, 88
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(b, t, ld, generator=g) * 3
    logits[0, 0, col0] = 0.0            # sigmoid(0) = 0.5 is NOT < 0.5 -> 1
    logits[0, 1, col0] = -1e-9          # rounds to 0.5 in fp32 -> 1, like the reference
    logits[0, 2, col0] = -1e-3
    ld_d = logits.cuda()
    out = torch.empty(b, keys, t, device='cuda')
    _lib.check(L.amtx_pianoroll_fwd(_lib.ptr(ld_d), ld, col0, b, t, keys, 0.5, _lib.ptr(out), _stream()))
    act = torch.sigmoid(logits[:, :, col0:col0 + keys]).transpose(-1, -2).contiguous()
    ref = act.clone()
    ref[ref < 0.5] = 0
    ref[ref != 0] = 1
    assert torch.equal(out.cpu(), ref)
    assert out[0, 0, 0] == 1 and out[0, 0, 1] == 1 and out[0, 0, 2] == 0
    _lib.check(L.amtx_pianoroll_fwd(_lib.ptr(ld_d), ld, col0, b, t, keys, -1.0, _lib.ptr(out), _stream()))
    assert (out.cpu() - act).abs().max().item() < 1e-6


_SKINNY_AB = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from amt_tools_amd import _lib
L = _lib.lib()
outs = {}
for m, n, k in ((2500, 88, 256), (1300, 88, 3648), (4099, 88, 512), (1024, 4, 128), (70000, 88, 1024)):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).cuda().to(torch.bfloat16).contiguous()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).numpy()
    bias = torch.randn(n, generator=g).cuda()
    packed = np.zeros(L.amtx_linear_packed_elems(n, k, 1), dtype=np.uint16)
    _lib.check(L.amtx_linear_pack(_lib.ptr(w), n, k, 1, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    c = torch.zeros(m, n + 4, device='cuda')
    _lib.check(L.amtx_linear_fwd(_lib.ptr(a), k, 0, _lib.ptr(wp), 1, _lib.ptr(bias), _lib.ptr(c), n + 4, 1, m, n, k, _lib.current_stream()))
    outs[f'{m}_{n}_{k}'] = c.cpu().numpy()
np.savez(sys.argv[1], **outs)
'''


def test_skinny_gemm_returns_the_bits_of_the_two_buffer_kernel(tmp_path):
    """Round 5: products with N <= 128 (the LogisticBanks, the folded pitch head) run on gemm_skinny_kernel (one block per CU, every wave
    streaming the A rows of its own 32-row slice through a private LDS ring two k-tiles ahead); AMTX_GEMM_NO_SKINNY=1 keeps
    gemm_glds_kernel<., 128>.  Same fragments, same k order: IDENTICAL outputs, ragged M, several row tiles per block (70 000 rows = 274 tiles on
    256 blocks), short and long K."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = {}
    for tag, extra in (('skinny', {}), ('glds', {'AMTX_GEMM_NO_SKINNY': '1'})):
        env = dict(os.environ)
        env.update(extra)
        files[tag] = str(tmp_path / f'{tag}.npz')
        subprocess.check_call([sys.executable, '-c', _SKINNY_AB, files[tag]], env=env, cwd=root)
    a, b = np.load(files['skinny']), np.load(files['glds'])
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 5
    for k in a.files:
        assert np.abs(a[k]).max() > 0
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
