#!/usr/bin/env python3
"""Which stages of the bf16 engine make its logit error?  CPU emulation (not collected by pytest; run by hand):
    python tests/precision_attribution.py [clips] [frames]
Restates the engine's bf16 data path in torch fp32 with ONE switch per stage: a stage that is "rounded" sees its input activations and
its (BatchNorm-folded) weights rounded to bf16 -- what the HIP kernels feed the matrix cores -- with fp32 accumulation; an unrounded
stage is the fp32 reference.  Prints, per stage, the error when ONLY that stage is rounded and when everything BUT that stage is rounded
(= what running that stage in the split-bf16 'x3' arithmetic would leave).  Reference: amt_tools/models/onsetsframes.py:94-136,375-463,
models/common.py:586-620 through oracle/model_ref.py (the fp32 restatement)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import model_ref, frontend_np as fe
from amt_tools_amd.synth import synth_state_dict, synth_clip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 625
FMT = os.environ.get('ATTR_FORMAT', 'bf16')            # 'bf16' or 'f16': the 16-bit operand format emulated
torch.set_num_threads(8)


def rnd(x):
    return x.bfloat16().float() if FMT == 'bf16' else x.half().float()


def fold(sd, prefix, l):
    w, b = sd[f'{prefix}.layer{l}.0.weight'], sd[f'{prefix}.layer{l}.0.bias']
    g, be = sd[f'{prefix}.layer{l}.1.weight'], sd[f'{prefix}.layer{l}.1.bias']
    mu, var = sd[f'{prefix}.layer{l}.1.running_mean'], sd[f'{prefix}.layer{l}.1.running_var']
    s = (g.double() / torch.sqrt(var.double() + 1e-5))
    return (w.double() * s[:, None, None, None]).float(), (be.double() + (b.double() - mu.double()) * s).float()


def conv_stage(x, w, shift, on):
    if on:
        x, w = rnd(x), rnd(w)
    return F.relu(F.conv2d(x, w, None, padding=1) + shift[None, :, None, None])


def lin(x, w, b, on):
    if on:
        x, w = rnd(x), rnd(w)
    return F.linear(x, w, b)


def lstm(xp, w_hh_f, w_hh_b, on):
    """xp (B,T,2,4H) = W_ih x + b_ih + b_hh; recurrence with h (and W_hh, xp) rounded when `on`."""
    Bn, Tn = xp.shape[:2]
    H = w_hh_f.shape[1]
    out = xp.new_zeros(Bn, Tn, 2 * H)
    if on:
        xp = rnd(xp)
    for d, whh in enumerate((w_hh_f, w_hh_b)):
        if on:
            whh = rnd(whh)
        h = xp.new_zeros(Bn, H)
        c = xp.new_zeros(Bn, H)
        for s in range(Tn):
            t = s if d == 0 else Tn - 1 - s
            gates = xp[:, t, d] + F.linear(rnd(h) if on else h, whh)
            i, f, g, o = gates.chunk(4, dim=-1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[:, t, H * d:H * (d + 1)] = h
    return out


def xproj(x, sd, p, on):
    w = torch.cat([sd[p + 'weight_ih_l0'], sd[p + 'weight_ih_l0_reverse']])
    b = torch.cat([sd[p + 'bias_ih_l0'] + sd[p + 'bias_hh_l0'], sd[p + 'bias_ih_l0_reverse'] + sd[p + 'bias_hh_l0_reverse']])
    return lin(x, w, b, on).reshape(x.shape[0], x.shape[1], 2, -1)


STAGES = ['conv1', 'conv2', 'conv3', 'fc1', 'pitch_head', 'rec_xproj', 'rec_lstm', 'rec_head', 'adj_xproj', 'adj_lstm', 'adj_head']


def forward(feats, sd, on):
    a3 = {}
    for head in ('onset_head.0', 'pitch_head.0'):
        x = feats
        for l, st in ((1, 'conv1'), (2, 'conv2'), (3, 'conv3')):
            w, sh = fold(sd, head, l)
            x = conv_stage(x, w, sh, st in on)
            if l > 1:
                x = F.max_pool2d(x, (1, 2))
        a3[head] = x.transpose(-3, -2).flatten(-2)
    e = lin(a3['onset_head.0'], sd['onset_head.0.fc1.0.weight'], sd['onset_head.0.fc1.0.bias'], 'fc1' in on)
    # pitch head: fc1 and LogisticBank folded into one layer, as the engine does
    wf = (sd['pitch_head.1.output_layer.weight'].double() @ sd['pitch_head.0.fc1.0.weight'].double()).float()
    bf = (sd['pitch_head.1.output_layer.weight'].double() @ sd['pitch_head.0.fc1.0.bias'].double() + sd['pitch_head.1.output_layer.bias'].double()).float()
    pitch = lin(a3['pitch_head.0'], wf, bf, 'pitch_head' in on)
    p = 'onset_head.1.mlm.'
    l1 = lstm(xproj(e, sd, p, 'rec_xproj' in on), sd[p + 'weight_hh_l0'], sd[p + 'weight_hh_l0_reverse'], 'rec_lstm' in on)
    onsets = lin(l1, sd['onset_head.2.output_layer.weight'], sd['onset_head.2.output_layer.bias'], 'rec_head' in on)
    joint = torch.cat((onsets, pitch), dim=-1)
    p = 'adjoin.0.mlm.'
    l2 = lstm(xproj(joint, sd, p, 'adj_xproj' in on), sd[p + 'weight_hh_l0'], sd[p + 'weight_hh_l0_reverse'], 'adj_lstm' in on)
    mp = lin(l2, sd['adjoin.1.output_layer.weight'], sd['adjoin.1.output_layer.bias'], 'adj_head' in on)
    return {'onsets': onsets, 'multi_pitch': mp, 'pitch_head': pitch}


def report(name, got, ref):
    row = [name]
    for k in ('onsets', 'multi_pitch'):
        dl = (got[k] - ref[k]).abs().max().item()
        ds = (torch.sigmoid(got[k]) - torch.sigmoid(ref[k])).abs().max().item()
        flips = ((got[k] >= 0) != (ref[k] >= 0)).float().mean().item()
        row.append(f'{k}: max|dlogit| {dl:.2e} max|dsigmoid| {ds:.2e} cells flipped {flips:.2e}')
    print(' | '.join(row), flush=True)


def main():
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=2).items()}
    feats = torch.from_numpy(np.stack([fe.melspec_process_audio(synth_clip(i, num_samples=512 * (T - 1)), 22050) for i in range(B)]).astype(np.float32))
    feats = feats.transpose(-1, -2).contiguous()       # process_audio gives (1, F, T) per clip -> (B, 1, T, F)
    with torch.no_grad():
        ref = forward(feats, sd, set())
        chk = model_ref.onsets_frames_logits(feats, sd)
        assert max((ref[k] - chk[k]).abs().max().item() for k in ref) < 2e-4, 'the unrounded emulation must be the oracle'
        print(f'{B} clips x {feats.shape[2]} frames, operand format {FMT}')
        report('all stages rounded     ', forward(feats, sd, set(STAGES)), ref)
        for s in STAGES:
            report(f'only {s:<18}', forward(feats, sd, {s}), ref)
        for s in STAGES:
            report(f'all but {s:<15}', forward(feats, sd, set(STAGES) - {s}), ref)
        for grp in (['conv1', 'conv2', 'conv3'], ['rec_lstm', 'adj_lstm'], ['rec_xproj', 'rec_lstm', 'rec_head', 'adj_xproj', 'adj_lstm', 'adj_head'],
                    ['fc1', 'pitch_head']):
            report('all but ' + '+'.join(grp), forward(feats, sd, set(STAGES) - set(grp)), ref)


if __name__ == '__main__':
    main()
