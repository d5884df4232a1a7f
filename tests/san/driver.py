"""Runs INSIDE the sanitizer process (LD_PRELOAD = clang's ASan runtime; started by tests/test_sanitized_host.py): drives the host halves
of the C ABI through tests/san/_build/libamtx_san.so -- weight packers at ragged sizes, amtx_of_model_create / set_tensor / finalize for
every engine configuration (their BatchNorm folding, fc1 permutation, fp64 head folding, fragment packing), the argument checks and
workspace carving of the forward entry points (the first kernel launch then reports "no device" through the shim: that is the expected
error), spectrogram and CQT plan builders.  numpy + ctypes only; any ASan / UBSan report aborts the process."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_lib = _load('amtx_lib_san', os.path.join(ROOT, 'amt_tools_amd', '_lib.py'))          # the product's own binding table, without the package
synth = _load('amtx_synth_san', os.path.join(ROOT, 'amt_tools_amd', 'synth.py'))
_lib.LIB_PATH = sys.argv[1]
L = _lib.lib()
P = _lib.ptr
missing = [s for s in _lib.declared_symbols() if not hasattr(L, s)]
assert not missing, missing
rng = np.random.default_rng(0)
n_calls = 0

# ---------------------------------------------------------------- weight packers (op-level C ABI)
for planes in (1, 2):
    for c_out in (32, 64):
        w = rng.standard_normal((c_out, 32, 3, 3)).astype(np.float32)
        sc = rng.random(c_out).astype(np.float32)
        out = np.zeros(L.amtx_conv3x3_packed_elems(c_out, planes), np.uint16)
        _lib.check(L.amtx_conv3x3_pack(P(w), P(sc), c_out, planes, P(out)))
        _lib.check(L.amtx_conv3x3_pack(P(w), None, c_out, planes, P(out)))
        n_calls += 2
    for c_in, c_out in ((48, 48), (48, 96), (32, 32), (16, 16)):
        n = L.amtx_conv3x3g_packed_elems(c_in, c_out, planes)
        if n > 0:
            w = rng.standard_normal((c_out, c_in, 3, 3)).astype(np.float32)
            out = np.zeros(n, np.uint16)
            _lib.check(L.amtx_conv3x3g_pack(P(w), None, c_in, c_out, planes, P(out)))
            n_calls += 1
    for n_, k_ in ((88, 256), (1024, 512), (512, 3648), (1024, 176), (88, 3648), (7, 5), (300, 1000), (2048, 72)):
        w = rng.standard_normal((n_, k_)).astype(np.float32)
        out = np.zeros(L.amtx_linear_packed_elems(n_, k_, planes), np.uint16)
        _lib.check(L.amtx_linear_pack(P(w), n_, k_, planes, P(out)))
        n_calls += 1
    for hid in (128, 256):
        wf = rng.standard_normal((4 * hid, hid)).astype(np.float32)
        wb = rng.standard_normal((4 * hid, hid)).astype(np.float32)
        out = np.zeros(L.amtx_bilstm_h_packed_elems(hid, planes), np.uint16)
        _lib.check(L.amtx_bilstm_h_pack(P(wf), P(wb), hid, planes, P(out)))
        n_calls += 1
    out = np.zeros(L.amtx_bilstm_packed_elems(planes), np.uint16)
    wf = rng.standard_normal((512, 128)).astype(np.float32)      # named: a pointer taken from a temporary would dangle
    wb = rng.standard_normal((512, 128)).astype(np.float32)
    _lib.check(L.amtx_bilstm_pack(P(wf), P(wb), planes, P(out)))
    n_calls += 1

# ---------------------------------------------------------------- engine: create / set_tensor / finalize / workspace / argument checks
configs = [dict(dim_in=229, ch=1, mc=2, off=0, prec=0), dict(dim_in=229, ch=1, mc=2, off=0, prec=1), dict(dim_in=229, ch=1, mc=2, off=0, prec=2),
           dict(dim_in=229, ch=1, mc=3, off=1, prec=0), dict(dim_in=229, ch=1, mc=3, off=0, prec=1), dict(dim_in=72, ch=6, mc=2, off=0, prec=0),
           dict(dim_in=72, ch=6, mc=3, off=1, prec=1), dict(dim_in=72, ch=6, mc=2, off=0, prec=2), dict(dim_in=229, ch=1, mc=3, off=1, prec=2), dict(dim_in=8, ch=1, mc=2, off=1, prec=0), dict(dim_in=40, ch=1, mc=2, off=0, prec=2),
           dict(dim_in=5, ch=1, mc=2, off=0, prec=0), dict(dim_in=192, ch=2, mc=2, off=0, prec=0)]
for cfg in configs:
    h = C.c_void_p()
    _lib.check(L.amtx_of_model_create(C.byref(h), cfg['dim_in'], cfg['ch'], cfg['mc'], 88, cfg['off'], cfg['prec']), 'create')
    sd = synth.synth_state_dict(3, dim_in=cfg['dim_in'], in_channels=cfg['ch'], model_complexity=cfg['mc'], offsets=bool(cfg['off']))
    keep = []
    for k, v in sd.items():
        a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
        if a.dtype.kind != 'f' or a.size == 0 or 'num_batches_tracked' in k:
            continue
        keep.append(a)
        _lib.check(L.amtx_of_model_set_tensor(h, k.encode(), P(a), a.size), 'set_tensor')
    _lib.check(L.amtx_of_model_finalize(h), 'finalize')
    for B, T in ((1, 1), (3, 17), (130, 47), (1024, 625)):
        need = L.amtx_of_workspace_bytes(h, B, T)
        assert need > 0
        L.amtx_of_conv_stack_fused(h, B, T)
    # forward: argument checks and workspace carving run on the host; the first kernel launch reports "no device" through the shim
    B, T = 2, 9
    need = L.amtx_of_workspace_bytes(h, B, T)
    ws = np.zeros(need + 256, np.uint8)
    base = (ws.ctypes.data + 255) // 256 * 256
    feats = rng.random((B, cfg['ch'], T, cfg['dim_in'])).astype(np.float32)
    on = np.zeros((B, 88, T), np.float32)
    mp = np.zeros((B, 88, T), np.float32)
    sb, sc_, st, sf = (s // 4 for s in feats.strides)
    rc = L.amtx_of_forward(h, P(feats), sb, sc_, st, sf, B, T, C.c_void_p(base), need, P(on), P(mp), None, None, None, None)
    assert rc != 0 and b'no device' in L.amtx_last_error(), (rc, L.amtx_last_error())
    rc = L.amtx_of_forward(h, P(feats), sb, sc_, st, sf, B, T, C.c_void_p(base), need - 1, P(on), P(mp), None, None, None, None)
    assert rc != 0 and b'workspace too small' in L.amtx_last_error()
    _lib.check(L.amtx_of_model_destroy(h))
    n_calls += 1
# an incomplete model must be refused by finalize, not read past a missing tensor
h = C.c_void_p()
_lib.check(L.amtx_of_model_create(C.byref(h), 229, 1, 2, 88, 0, 0))
assert L.amtx_of_model_finalize(h) != 0
_lib.check(L.amtx_of_model_destroy(h))

# ---------------------------------------------------------------- spectrogram and CQT plans (host-built tables, uploaded through the shim)
for sr, n_fft, hop, win, n_mels, htk, center, pad in ((22050, 2048, 512, 2048, 229, 0, 1, 0), (22050, 2048, 512, 2048, 229, 1, 1, 1), (16000, 1024, 160, 800, 80, 0, 0, 0),
                                                     (22050, 4096, 512, 4096, 512, 1, 1, 0), (22050, 128, 64, 128, 40, 0, 1, 0), (44100, 2048, 441, 2048, 0, 0, 1, 0)):
    pl = C.c_void_p()
    rc = L.amtx_spec_plan_create(C.byref(pl), sr, n_fft, hop, win, n_mels, htk, center, pad)
    if rc != 0:
        continue
    nb = L.amtx_spec_num_bins(pl)
    assert nb > 0 and L.amtx_spec_num_frames(pl, 22050) > 0
    if n_mels:
        fb = np.zeros((n_mels, n_fft // 2 + 1), np.float32)
        _lib.check(L.amtx_spec_filterbank(pl, P(fb)))
        assert np.isfinite(fb).all() and fb.max() > 0
    _lib.check(L.amtx_spec_plan_destroy(pl))
    n_calls += 1
    if n_mels:       # the slot matching of the mel gather (mel_assign_slots), host-only
        srow, sstart, rmax = np.full(512, -5, np.int32), np.zeros(512, np.int32), np.zeros(8, np.int32)
        rounds = L.amtx_spec_mel_layout(sr, n_fft, n_mels, htk, P(srow), P(sstart), P(rmax))
        assert rounds == (n_mels + 63) // 64 and sorted(srow[:64 * rounds][srow[:64 * rounds] >= 0].tolist()) == list(range(n_mels))
        n_calls += 1
for fmin, n_bins, bpo, harm in ((82.41, 192, 24, [1.0]), (32.70, 72, 12, [0.5, 1, 2, 3, 4, 5]), (27.5, 88, 12, [1.0])):
    pl = C.c_void_p()
    hv = (C.c_double * len(harm))(*harm)
    rc = L.amtx_cqt_plan_create(C.byref(pl), 22050, 512, fmin, n_bins, bpo, 0.0, hv, len(harm), 1, 0)
    if rc == 0:
        assert L.amtx_cqt_num_harmonics(pl) == len(harm) and L.amtx_cqt_num_frames(pl, 22050 * 4) > 0
        assert L.amtx_cqt_workspace_bytes(pl, 3, 22050 * 4) > 0
        _lib.check(L.amtx_cqt_plan_destroy(pl))
        n_calls += 1
print(f'sanitized host halves: {n_calls} packer / model / plan exercises, no report')
