"""
The training step's layers as autograd.Functions over hand-written HIP forward + backward kernels.

The reference trains through nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.LSTM (amt_tools/models/onsetsframes.py:330-575 inside
amt_tools/train.py:126-141).  On ROCm the LSTM is MIOpen's per-time-step kernel chain (thousands of tiny launches per step, 46 of
the 61 ms of GPU time of a step at 8 clips x 625 frames).  Here:
  * BiLSTMFunction: the T dependent steps of both directions are ONE persistent kernel forward (amtx_bilstm_h_train_fwd, gates and
    cell states saved) and ONE backward (amtx_bilstm_h_train_bwd -> dL/d(xproj)); the input projection and every parameter gradient
    are GEMMs over B*T on csrc/train.hip (matmul_f32);
  * Conv3x3Function / LinearFunction: forward, input and weight gradients as split-bf16 implicit GEMMs (csrc/train.hip);
  * BNReLUPoolFunction (csrc/bn.hip), BCELogitsLossFunction (csrc/head.hip).
Arithmetic: fp32 in / out with split-bf16 (3-MFMA) products, fp32 accumulation -- fp32-class accuracy.  No vendor BLAS / MIOpen kernel
runs in a step; a layer whose shape the kernels do not take goes to the stock ATen op AND is recorded (`fallbacks()`,
`training_backend()`), or raises under `AMTX_STRICT_TRAINING=1`.
"""
import os

import torch

from . import _lib

__all__ = ['bilstm', 'bilstm_multi', 'BiLSTMFunction', 'bce_logits_loss', 'BCELogitsLossFunction', 'bn_relu_pool', 'BNReLUPoolFunction',
           'matmul_f32', 'linear', 'LinearFunction', 'conv3x3', 'Conv3x3Function', 'training_backend', 'note_fallback', 'fallbacks', 'fallback_total',
           'reset_fallbacks']

HIDDEN_SIZES = (128, 256, 384, 512)  # hidden sizes per direction the training recurrences are built for (model_complexity 2 .. 5)


USE_HIP_DENSE = True            # False: nn.Conv2d / nn.Linear / the LSTM's matmuls through ATen (MIOpen / hipBLASLt) -- the A/B switch of the tests


_FALLBACKS = {}                 # site -> [reason, count]: layers of a GPU training step that went to a stock ATen op
_FALLBACK_TOTAL = 0             # every fallback ever recorded (never reset): a model brackets its own forward with it


def note_fallback(site, reason):
    """A layer of the GPU training path is about to run on the stock ATen op (MIOpen / hipBLASLt) instead of the HIP kernels: say so.
    `AMTX_STRICT_TRAINING=1` turns it into an error -- the inference engine never falls back, training only does so on the record."""
    if os.environ.get('AMTX_STRICT_TRAINING', '0') not in ('', '0'):
        raise RuntimeError(f'amt_tools_amd training path: {site} would fall back to ATen ({reason}) and AMTX_STRICT_TRAINING is set')
    global _FALLBACK_TOTAL
    rec = _FALLBACKS.setdefault(site, [reason, 0])
    rec[0] = reason
    rec[1] += 1
    _FALLBACK_TOTAL += 1


def fallbacks():
    """{site: (reason, times taken)} since the last reset_fallbacks()."""
    return {k: (v[0], v[1]) for k, v in _FALLBACKS.items()}


def fallback_total():
    """Monotonic count of recorded fallbacks (reset_fallbacks() does not touch it): the difference across a model's own training forward
    says whether THAT forward took one, whatever other models or earlier validation passes of the process recorded."""
    return _FALLBACK_TOTAL


def reset_fallbacks():
    _FALLBACKS.clear()


def training_backend():
    """One line for benchmark records: which kernels the GPU training steps run so far REALLY ran on (the switches' nominal state plus
    every recorded ATen fallback)."""
    if USE_HIP_DENSE:
        line = ('HIP: Conv2d fwd/dgrad/wgrad and every Linear / LSTM projection fwd/bwd (split-bf16 implicit GEMMs), BiLSTM fwd/bwd, '
                'BatchNorm+ReLU+MaxPool fwd/bwd, BCE loss+grad; ATen: elementwise glue, Dropout, Adam')
    else:
        line = ('HIP: BiLSTM fwd/bwd, BatchNorm+ReLU+MaxPool fwd/bwd, BCE loss+grad; ATen (MIOpen / hipBLASLt): conv fwd/dgrad/wgrad, '
                'Linear fwd/bwd, Adam')
    if _FALLBACKS:
        line += '; ATen FALLBACKS TAKEN: ' + '; '.join(f'{k} ({v[0]}, x{v[1]})' for k, v in sorted(_FALLBACKS.items()))
    return line


# ---------------------------------------------------------------------------------------------------------------------------
# dense layers of the training step on the amtx_matmul_f32 / amtx_linear_* / amtx_conv3x3_* kernels (csrc/train.hip)
# ---------------------------------------------------------------------------------------------------------------------------
_WS = {}


def _workspace(nbytes, device):
    """Grow-only scratch buffer per device and stream (split-contraction partials, permuted conv weights).  Stream-ordered reuse: every
    consumer is enqueued on the current stream right behind its producer."""
    # one buffer per (device, stream): the reuse is only ordered within a stream, so two streams (or two threads on their own streams)
    # training on one GPU each get their own scratch instead of silently sharing split-contraction partials
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = _lib.alloc_workspace(max(int(nbytes), 1 << 20), device)
    return ws


def _ok2d(t):
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def matmul_f32(a, b, a_trans=False, b_trans=False, bias=None, out=None):
    """C[m, n] = sum_k A(m, k) B(n, k) (+ bias[n]) on the HIP kernel, no autograd.  `a` is the (m, k) matrix (a_trans False) or is
    stored as (k, m) (a_trans True); `b` likewise with n.  Rows may be strided views (e.g. half of a concatenated buffer)."""
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and _ok2d(a) and _ok2d(b), 'matmul_f32: unsupported operand layout'
    m, k = (a.shape[1], a.shape[0]) if a_trans else a.shape
    n, k2 = (b.shape[1], b.shape[0]) if b_trans else b.shape
    assert k == k2
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    L = _lib.lib()
    need = int(L.amtx_matmul_workspace_bytes(m, n, k))
    ws = _workspace(need, a.device) if need else None
    with torch.cuda.device(a.device):
        _lib.check(L.amtx_matmul_f32(_lib.ptr(a), a.stride(0), int(a_trans), _lib.ptr(b), b.stride(0), int(b_trans), _lib.ptr(bias), _lib.ptr(out),
                                     out.stride(0), m, n, k, _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.current_stream(a.device)),
                   'amtx_matmul_f32')
    return out


def _colsum(x2):
    """Column sums of a (rows, n) matrix (bias gradients) through amtx_linear_bwd's db output."""
    m, n = x2.shape
    L = _lib.lib()
    db = torch.empty(n, dtype=torch.float32, device=x2.device)
    ws = _workspace(int(L.amtx_linear_bwd_workspace_bytes(m, n, 4)), x2.device)
    with torch.cuda.device(x2.device):
        _lib.check(L.amtx_linear_bwd(_lib.ptr(x2), x2.stride(0), None, 0, None, 0, None, 0, None, _lib.ptr(db), m, n, 4, _lib.ptr(ws), ws.numel(),
                                     _lib.current_stream(x2.device)), 'amtx_linear_bwd')
    return db


def linear_supported(x, weight):
    return (USE_HIP_DENSE and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2
            and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0 and x.shape[-1] == weight.shape[1])


class LinearFunction(torch.autograd.Function):
    """nn.Linear (fc1, LogisticBank.output_layer): y = x w^T + b on amtx_linear_train_fwd, gradients on amtx_linear_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        if not _ok2d(x2):
            x2 = x2.contiguous()
        w = weight if _ok2d(weight) else weight.contiguous()
        y = matmul_f32(x2, w, bias=bias)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.shape = x.shape
        return y.reshape(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        n, k = w.shape
        dy2 = dy.reshape(-1, n)
        if not _ok2d(dy2):
            dy2 = dy2.contiguous()
        m = dy2.shape[0]
        L = _lib.lib()
        dev = dy2.device
        dx = torch.empty((m, k), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty((n, k), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        db = torch.empty(n, dtype=torch.float32, device=dev) if ctx.has_bias and ctx.needs_input_grad[2] else None
        ws = _workspace(int(L.amtx_linear_bwd_workspace_bytes(m, n, k)), dev)
        with torch.cuda.device(dev):
            _lib.check(L.amtx_linear_bwd(_lib.ptr(dy2), dy2.stride(0), _lib.ptr(x2), x2.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(dx),
                                         k, _lib.ptr(dw), _lib.ptr(db), m, n, k, _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)),
                       'amtx_linear_bwd')
        return (dx.reshape(ctx.shape) if dx is not None else None), dw, db


def linear(x, weight, bias=None):
    """F.linear on the HIP kernels, differentiably (fp32 CUDA tensors; in/out features multiples of 4)."""
    return LinearFunction.apply(x, weight, bias)


def conv3x3_supported(x, conv):
    """nn.Conv2d(3x3, padding 1, stride 1) on an fp32 CUDA map whose input gradient, if wanted, the kernels can produce."""
    if not (USE_HIP_DENSE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and isinstance(conv, torch.nn.Conv2d)):
        return False
    if not (conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.bias is not None and conv.padding_mode == 'zeros'):
        return False
    c_in, c_out = conv.in_channels, conv.out_channels
    if c_in == 1 and not x.requires_grad:
        return c_out % 8 == 0                 # direct first-layer kernel forward, K = 9 weight-gradient GEMM
    return c_out % 4 == 0                     # c_in a multiple of 4, or padded to one with zero channels (conv3x3)


class Conv3x3Function(torch.autograd.Function):
    """nn.Conv2d(c_in, c_out, 3, padding 1) on channels-last fp32 maps: implicit GEMMs forward (amtx_conv3x3_train_fwd), input and weight
    gradients (amtx_conv3x3_bwd).  x (B, c_in, T, F) -> y (B, c_out, T, F), both in channels-last memory format."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, Ci, T, F = x.shape
        Co = weight.shape[0]
        L = _lib.lib()
        xc = x.contiguous(memory_format=torch.channels_last) if Ci > 1 else x.contiguous()
        w = weight.contiguous()
        y = torch.empty((B, Co, T, F), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        ws = _workspace(int(L.amtx_conv3x3_train_workspace_bytes(B * T, F, Ci, Co)), x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.amtx_conv3x3_train_fwd(_lib.ptr(xc), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), B * T, T, F, Ci, Co, _lib.ptr(ws), ws.numel(),
                                                _lib.current_stream(x.device)), 'amtx_conv3x3_train_fwd')
        ctx.save_for_backward(xc, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w = ctx.saved_tensors
        B, Ci, T, F = xc.shape
        Co = w.shape[0]
        L = _lib.lib()
        dev = dy.device
        dyc = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, Ci, T, F), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(Co, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        ws = _workspace(int(L.amtx_conv3x3_train_workspace_bytes(B * T, F, Ci, Co)), dev)
        with torch.cuda.device(dev):
            _lib.check(L.amtx_conv3x3_bwd(_lib.ptr(dyc), _lib.ptr(xc), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), B * T, T, F, Ci, Co,
                                          _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)), 'amtx_conv3x3_bwd')
        return dx, dw, db


def conv3x3(x, conv):
    """`conv(x)` for an nn.Conv2d(3x3, padding 1) on the HIP kernels, differentiably.  Input channel counts the implicit GEMMs do not
    take as they are (3 or 6 = an HCQT / multi-channel first layer, amt_tools/features/hvqt.py:107-133; 1 when the input itself needs a
    gradient) are padded with zero channels to a multiple of 4 -- in the map and in the weights, so the products are unchanged and the
    weight gradient of the real channels comes back through the pad's own backward (a slice)."""
    c_in = x.shape[1]
    weight = conv.weight
    if c_in % 4 != 0 and not (c_in == 1 and not x.requires_grad):
        pad = (-c_in) % 4
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, pad))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, pad))
    return Conv3x3Function.apply(x, weight, conv.bias)


def _pack(w_hh_f, w_hh_b):
    L = _lib.lib()
    H = w_hh_f.shape[1]
    n = int(L.amtx_bilstm_h_packed_elems(H, 2))
    fwd = torch.empty(n, dtype=torch.int16, device=w_hh_f.device)
    bwd = torch.empty(n, dtype=torch.int16, device=w_hh_f.device)
    wf, wb = w_hh_f.detach().contiguous().float(), w_hh_b.detach().contiguous().float()
    _lib.check(L.amtx_bilstm_h_pack_device(_lib.ptr(wf), _lib.ptr(wb), H, 2, _lib.ptr(fwd), _lib.ptr(bwd), _lib.current_stream(wf.device)),
               'amtx_bilstm_h_pack_device')
    return fwd, bwd


class BiLSTMFunction(torch.autograd.Function):
    """y_g = BiLSTM_g(x_g) for G independent LSTMs of the same (B, T, hidden) in ONE launch per direction of time (forward kernel,
    backward kernel), PyTorch's parameter layout (gate order i, f, g, o; weight_ih (4H, I_g), weight_hh (4H, H), two biases per
    direction), zero initial state, batch_first, H in HIDDEN_SIZES.  Arguments: G, then per LSTM x (B, T, I_g) fp32 CUDA and its
    eight parameters; returns G tensors (B, T, 2H)."""

    @staticmethod
    def forward(ctx, G, *args):
        assert len(args) == 9 * G
        groups = [args[9 * g:9 * g + 9] for g in range(G)]
        x0 = groups[0][0]
        B, T = x0.shape[:2]
        H = groups[0][2].shape[1]
        L = _lib.lib()
        dev = x0.device
        n = int(L.amtx_bilstm_h_packed_elems(H, 2))
        xproj = torch.empty((G, B * T, 8 * H), dtype=torch.float32, device=dev)
        frag_fwd = torch.empty((G, n), dtype=torch.int16, device=dev)
        frag_bwd = torch.empty((G, n), dtype=torch.int16, device=dev)
        saved = []
        for g, (x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_b, w_hh_b, b_ih_b, b_hh_b) in enumerate(groups):
            assert x.shape[:2] == (B, T) and w_hh_f.shape[1] == H
            x2 = x.reshape(B * T, x.shape[2])
            w_ih = torch.cat([w_ih_f, w_ih_b], dim=0)                                  # (8 H, I)
            bias = torch.cat([b_ih_f + b_hh_f, b_ih_b + b_hh_b], dim=0)
            hip_mm = USE_HIP_DENSE and x2.shape[1] % 4 == 0
            if not hip_mm:
                note_fallback('BiLSTM input projection', 'USE_HIP_DENSE off' if not USE_HIP_DENSE else f'input size {x2.shape[1]} is not a multiple of 4')
            if hip_mm:
                if not _ok2d(x2):
                    x2 = x2.contiguous()
                matmul_f32(x2, w_ih, bias=bias, out=xproj[g])                          # [B][T][2][4H]
            else:
                torch.addmm(bias, x2, w_ih.t(), out=xproj[g])
            wf, wb = w_hh_f.detach().contiguous().float(), w_hh_b.detach().contiguous().float()
            _lib.check(L.amtx_bilstm_h_pack_device(_lib.ptr(wf), _lib.ptr(wb), H, 2, _lib.ptr(frag_fwd[g]), _lib.ptr(frag_bwd[g]),
                                                   _lib.current_stream(dev)), 'amtx_bilstm_h_pack_device')
            saved += [x2, w_ih]
        out = torch.empty((G, B, T, 2 * H), dtype=torch.float32, device=dev)
        save = torch.empty((G, B, T, 2, 5, H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.amtx_bilstm_h_train_fwd(_lib.ptr(xproj), _lib.ptr(frag_fwd), H, 2, _lib.ptr(out), _lib.ptr(save), B, T, G,
                                                 _lib.current_stream(dev)), 'amtx_bilstm_h_train_fwd')
        ctx.save_for_backward(out, save, frag_bwd, *saved)
        ctx.hip_mm = [USE_HIP_DENSE and saved[2 * g].shape[1] % 4 == 0 for g in range(G)]
        ctx.dims = (G, B, T, H)
        ctx.need_dx = [bool(ctx.needs_input_grad[1 + 9 * g]) for g in range(G)]
        return tuple(out[g] for g in range(G))

    @staticmethod
    def backward(ctx, *douts):
        out, save, frag_bwd, *saved = ctx.saved_tensors
        G, B, T, H = ctx.dims
        L = _lib.lib()
        dev = out.device
        dout = torch.stack([d.contiguous().float() for d in douts], dim=0)
        dxproj = torch.empty((G, B * T, 8 * H), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.amtx_bilstm_h_train_bwd(_lib.ptr(dout), _lib.ptr(save), _lib.ptr(frag_bwd), H, 2, _lib.ptr(dxproj), B, T, G,
                                                 _lib.current_stream(dev)), 'amtx_bilstm_h_train_bwd')
        ones = torch.ones((1, B * T), dtype=torch.float32, device=dev)
        grads = [None]
        n = 4 * H
        for g in range(G):
            x2, w_ih = saved[2 * g], saved[2 * g + 1]
            dxp = dxproj[g]
            # h_{t-1} of the forward direction / h_{t+1} of the backward direction (zero initial state)
            hp_f = torch.zeros((B, T, H), dtype=torch.float32, device=dev)
            hp_f[:, 1:] = out[g, :, :-1, :H]
            hp_b = torch.zeros((B, T, H), dtype=torch.float32, device=dev)
            hp_b[:, :-1] = out[g, :, 1:, H:]
            if ctx.hip_mm[g]:
                dx = matmul_f32(dxp, w_ih, b_trans=True).reshape(B, T, x2.shape[1]) if ctx.need_dx[g] else None     # dxp @ w_ih
                dw_ih = matmul_f32(dxp, x2, a_trans=True, b_trans=True)                                            # dxp^T @ x2, (8 H, I)
                db = _colsum(dxp)
                dw_hh_f = matmul_f32(dxp[:, :n], hp_f.reshape(B * T, H), a_trans=True, b_trans=True)               # (4 H, H)
                dw_hh_b = matmul_f32(dxp[:, n:], hp_b.reshape(B * T, H), a_trans=True, b_trans=True)
            else:
                dx = (dxp @ w_ih).reshape(B, T, x2.shape[1]) if ctx.need_dx[g] else None
                dw_ih = dxp.t() @ x2                                                  # (8 H, I)
                # column sums as a (1, B*T) x (B*T, 8 H) product: ATen's reduce kernel over the strided dimension is slower
                db = ones.mm(dxp).squeeze(0)
                dg = dxp.reshape(B * T, 2, 4 * H)
                dw_hh_f = dg[:, 0].t() @ hp_f.reshape(B * T, H)
                dw_hh_b = dg[:, 1].t() @ hp_b.reshape(B * T, H)
            grads += [dx, dw_ih[:n], dw_hh_f, db[:n], db[:n], dw_ih[n:], dw_hh_b, db[n:], db[n:]]
        return tuple(grads)


def _lstm_args(x, lstm):
    return (x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
            lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)


def bilstm(x, lstm):
    """Run `lstm` (an nn.LSTM(batch_first, bidirectional, hidden in HIDDEN_SIZES, one layer)) on x through the HIP kernels, differentiably."""
    return BiLSTMFunction.apply(1, *_lstm_args(x, lstm))[0]


def bilstm_multi(xs, lstms):
    """Independent LSTMs of the same hidden size on inputs of the same (B, T) in one launch each way (the onset and offset
    recurrences of OnsetsFrames2, which otherwise run one after the other with two blocks busy each)."""
    args = []
    for x, lstm in zip(xs, lstms):
        args += list(_lstm_args(x, lstm))
    return BiLSTMFunction.apply(len(xs), *args)


class BCELogitsLossFunction(torch.autograd.Function):
    """LogisticBank.get_loss (amt_tools/models/common.py:541-584): logits (B,T,K), labels (B,K,T), optional per-key weights ->
    mean_b sum_k mean_t BCEWithLogits, with d loss / d logits produced by the same kernel pass (amtx_bce_logits_loss)."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        B, T, K = logits.shape
        L = _lib.lib()
        x = logits.detach().contiguous().float()
        y = labels.detach().contiguous().float()
        w = weight.detach().contiguous().float() if weight is not None else None
        need_grad = logits.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty((B, T, K), dtype=torch.float32, device=x.device) if need_grad else None
        ws = torch.empty(int(L.amtx_bce_logits_loss_workspace_bytes(B, T, K)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.amtx_bce_logits_loss(_lib.ptr(x), K, _lib.ptr(y), _lib.ptr(w), B, T, K, _lib.ptr(loss), _lib.ptr(grad),
                                              _lib.ptr(ws), ws.numel(), _lib.current_stream(x.device)), 'amtx_bce_logits_loss')
        # saved through autograd (not as a plain attribute of ctx): the (B,T,K) gradient is then released with the graph when backward never
        # runs (a labelled forward under enable_grad whose loss is only read)
        ctx.save_for_backward(*([grad] if grad is not None else []))
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        return (saved[0] * g if saved else None), None, None


def bce_logits_loss(logits, labels, weight=None):
    """(B,T,K) fp32 CUDA logits, (B,K,T) labels -> scalar loss attached to the autograd graph of `logits`."""
    return BCELogitsLossFunction.apply(logits, labels, weight)


class BNReLUPoolFunction(torch.autograd.Function):
    """BatchNorm2d with batch statistics + ReLU (+ MaxPool2d((1, 2))) of the acoustic model's conv stages
    (amt_tools/models/onsetsframes.py:375-416) as two HIP passes forward and two backward (amtx_bn_relu_pool_train_fwd / _bwd)
    instead of MIOpen BatchNorm + elementwise ReLU / pooling kernels.  x (B, C, T, F) fp32 in channels-last memory format."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, pool):
        B, Cc, T, F = x.shape
        L = _lib.lib()
        Fo = F // 2 if pool else F
        y = torch.empty((B, Cc, T, Fo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        stats = torch.empty((4, Cc), dtype=torch.float32, device=x.device)
        ws = torch.empty(int(L.amtx_bn_train_workspace_bytes(Cc)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.amtx_bn_relu_pool_train_fwd(_lib.ptr(x), B * T, F, Cc, int(pool), _lib.ptr(weight), _lib.ptr(bias), float(eps),
                                                     float(momentum), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(y),
                                                     _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.current_stream(x.device)),
                       'amtx_bn_relu_pool_train_fwd')
        ctx.save_for_backward(x, stats)
        ctx.pool = int(pool)
        ctx.ws = ws
        ctx.affine = weight is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        B, Cc, T, F = x.shape
        L = _lib.lib()
        dy = dy.contiguous(memory_format=torch.channels_last).float()
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device) if ctx.affine else None
        dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device) if ctx.affine else None
        with torch.cuda.device(x.device):
            _lib.check(L.amtx_bn_relu_pool_train_bwd(_lib.ptr(x), B * T, F, Cc, ctx.pool, _lib.ptr(stats), _lib.ptr(dy), _lib.ptr(dx),
                                                     _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ctx.ws), ctx.ws.numel(),
                                                     _lib.current_stream(x.device)), 'amtx_bn_relu_pool_train_bwd')
        return dx, dgamma, dbeta, None, None, None, None, None


def bn_relu_pool_supported(x, bn):
    """The HIP path takes fp32 channels-last CUDA maps in training mode with a fixed momentum and tracked running statistics."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.training and bn.track_running_stats and bn.momentum is not None
            and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and x.is_contiguous(memory_format=torch.channels_last))


def bn_relu_pool(x, bn, pool):
    """nn.Sequential(bn, ReLU(), [MaxPool2d((1, 2))]) on x through the HIP kernels, differentiably, with bn's running statistics
    and num_batches_tracked updated as nn.BatchNorm2d does in training mode."""
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return BNReLUPoolFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, bool(pool))
