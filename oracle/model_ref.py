"""
ORACLE (test infrastructure only) -- plain fp32 CPU restatement of the
Onsets & Frames model half of amt-tools' hot path, written against
`state_dict` tensors so it shares no module code with the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (amt_tools_amd.*) never does.

What it restates (all citations relative to /root/reference):
* amt_tools/models/onsetsframes.py:330-463  AcousticModel  (conv3x3+BN+ReLU x3, MaxPool(1,2) x2, fc1)
* amt_tools/models/onsetsframes.py:466-575  LanguageModel  (bidirectional nn.LSTM; eval chunking is a numerical no-op)
* amt_tools/models/onsetsframes.py:94-136   OnsetsFrames.forward  (pitch head, onset head, cat -> adjoin)
* amt_tools/models/onsetsframes.py:235-282  OnsetsFrames2.forward (adds the offset head)
* amt_tools/models/common.py:541-584        LogisticBank.get_loss (BCE with logits: mean_T, sum_keys, mean_B)
* amt_tools/models/common.py:586-620        LogisticBank.finalize_output (sigmoid, transpose, threshold)
* amt_tools/tools/utils.py:2896-2919        threshold_activations
* amt_tools/tools/utils.py:2381-2412        multi_pitch_to_onsets

Pinned against golden vectors generated in the build container from the real
reference classes (tools/gen_golden.py -> tests/golden/of1_*.npz); see
tests/test_oracle_model.py.
"""

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, sd, prefix, training):
    """nn.BatchNorm2d: batch statistics in training mode, running statistics in eval mode."""
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    if training:
        return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], w, b, False, 0.0, BN_EPS)


def acoustic_model(feats, sd, prefix, training=False):
    """AcousticModel.forward (onsetsframes.py:432-463).  feats (B,C,T,F) -> (B,T,dim_out).  Dropout is
    the identity in eval mode; the training-mode oracle is only used with dropout disabled (p=0)."""
    x = F.conv2d(feats, sd[prefix + '.layer1.0.weight'], sd[prefix + '.layer1.0.bias'], padding=1)
    x = F.relu(_bn(x, sd, prefix + '.layer1.1', training))
    x = F.conv2d(x, sd[prefix + '.layer2.0.weight'], sd[prefix + '.layer2.0.bias'], padding=1)
    x = F.max_pool2d(F.relu(_bn(x, sd, prefix + '.layer2.1', training)), (1, 2))
    x = F.conv2d(x, sd[prefix + '.layer3.0.weight'], sd[prefix + '.layer3.0.bias'], padding=1)
    x = F.max_pool2d(F.relu(_bn(x, sd, prefix + '.layer3.1', training)), (1, 2))
    x = x.transpose(-3, -2).flatten(-2)
    return F.linear(x, sd[prefix + '.fc1.0.weight'], sd[prefix + '.fc1.0.bias'])


def _lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.LSTM, explicit time loop, PyTorch gate order i,f,g,o."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gates = F.linear(x[:, t], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, g, o = gates.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


LSTM_IMPL = 'loop'      # 'loop': the explicit time loop above (the restatement proper); 'aten': the same weights through
                        # torch.nn.LSTM, i.e. the ATen kernel the reference itself runs on a CPU -- used by bench.py's
                        # cpu_baseline so that the baseline is not slowed down by 2 x T Python iterations per recurrence
                        # (tests/test_oracle_model.py checks that both give the same values)


def _lstm_aten(x, sd, p):
    """The same weights through ATen's fused CPU LSTM (what nn.LSTM.forward calls: torch._VF.lstm), functional form: differentiable with
    respect to the `sd` tensors, so the full-size training-step parity test (8 clips x 625 frames, every parameter gradient) can use it."""
    H = sd[p + 'weight_hh_l0'].shape[1]
    names = ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse',
             'bias_ih_l0_reverse', 'bias_hh_l0_reverse')
    z = x.new_zeros(2, x.shape[0], H)
    return torch._VF.lstm(x, (z, z), [sd[p + n] for n in names], True, 1, 0.0, False, True, True)[0]


def language_model(x, sd, prefix):
    """LanguageModel.forward (onsetsframes.py:504-575): full-sequence BiLSTM, fwd | bwd concatenated."""
    p = prefix + '.mlm.'
    if LSTM_IMPL == 'aten':
        return _lstm_aten(x, sd, p)
    fwd = _lstm_direction(x, sd[p + 'weight_ih_l0'], sd[p + 'weight_hh_l0'],
                          sd[p + 'bias_ih_l0'], sd[p + 'bias_hh_l0'], False)
    bwd = _lstm_direction(x, sd[p + 'weight_ih_l0_reverse'], sd[p + 'weight_hh_l0_reverse'],
                          sd[p + 'bias_ih_l0_reverse'], sd[p + 'bias_hh_l0_reverse'], True)
    return torch.cat((fwd, bwd), dim=-1)


def logistic_bank(x, sd, prefix):
    return F.linear(x, sd[prefix + '.output_layer.weight'], sd[prefix + '.output_layer.bias'])


def onsets_frames_logits(feats, sd, training=False, detach_heads=False):
    """OnsetsFrames.forward / OnsetsFrames2.forward restated.  feats (B,C,T,F) (i.e. after the
    transpose of pre_proc, onsetsframes.py:90).  Returns dict of raw logits (B,T,88) plus the
    intermediate pitch-head logits under 'pitch_head'."""
    has_offsets = any(k.startswith('offset_head.') for k in sd)
    out = {}
    multi_pitch = logistic_bank(acoustic_model(feats, sd, 'pitch_head.0', training), sd, 'pitch_head.1')
    onsets = logistic_bank(language_model(acoustic_model(feats, sd, 'onset_head.0', training), sd, 'onset_head.1'),
                           sd, 'onset_head.2')
    out['onsets'] = onsets
    out['pitch_head'] = multi_pitch
    parts = [onsets.detach() if detach_heads else onsets]
    if has_offsets:
        offsets = logistic_bank(language_model(acoustic_model(feats, sd, 'offset_head.0', training), sd,
                                               'offset_head.1'), sd, 'offset_head.2')
        out['offsets'] = offsets
        parts.append(offsets.detach() if detach_heads else offsets)
    parts.append(multi_pitch)
    joint = torch.cat(parts, dim=-1)
    out['multi_pitch'] = logistic_bank(language_model(joint, sd, 'adjoin.0'), sd, 'adjoin.1')
    return out


def bce_loss(logits, reference):
    """LogisticBank.get_loss (models/common.py:541-584): logits (B,T,O), reference (B,O,T)."""
    est = logits.transpose(-2, -1)
    loss = F.binary_cross_entropy_with_logits(est.float(), reference.float(), reduction='none')
    return loss.mean(dim=-1).sum(dim=-1).mean()


def finalize(logits, threshold=0.5):
    """LogisticBank.finalize_output: sigmoid -> (B,O,T) -> optional threshold to {0,1}."""
    act = torch.sigmoid(logits.detach().clone()).transpose(-2, -1).contiguous()
    if threshold is not None:
        act[act < threshold] = 0
        act[act != 0] = 1
    return act


def multi_pitch_to_onsets(multi_pitch):
    """tools/utils.py:2381-2412 (works on ndarray or tensor-as-ndarray)."""
    mp = np.asarray(multi_pitch)
    first = mp[..., :1]
    diff = mp[..., 1:] - mp[..., :-1]
    onsets = np.concatenate([first, diff], axis=-1)
    onsets[onsets <= 0] = 0
    return onsets


def run_on_batch(feats_bcft, sd, labels=None, training=False, detach_heads=False):
    """TranscriptionModel.run_on_batch for OnsetsFrames(2) restated: features (B,C,F,T) as the datasets
    hand them over; optional labels dict with 'multi_pitch' (and 'onsets', 'offsets') of shape (B,88,T)."""
    feats = feats_bcft.transpose(-1, -2)
    logits = onsets_frames_logits(feats, sd, training, detach_heads)
    out = {'logits': logits}
    if labels is not None and 'multi_pitch' in labels:
        loss = {}
        loss['loss_pitch'] = bce_loss(logits['multi_pitch'], labels['multi_pitch'])
        onsets_ref = labels['onsets'] if 'onsets' in labels else \
            torch.from_numpy(multi_pitch_to_onsets(labels['multi_pitch'].numpy()))
        loss['loss_onsets'] = bce_loss(logits['onsets'], onsets_ref)
        loss['loss_total'] = loss['loss_pitch'] + loss['loss_onsets']
        if 'offsets' in logits:
            loss['loss_offsets'] = bce_loss(logits['offsets'], labels['offsets'])
            loss['loss_total'] = loss['loss_total'] + loss['loss_offsets']
        out['loss'] = loss
    out['onsets'] = finalize(logits['onsets'], 0.5)
    out['multi_pitch'] = finalize(logits['multi_pitch'], 0.5)
    if 'offsets' in logits:
        out['offsets'] = finalize(logits['offsets'], None)
    return out
