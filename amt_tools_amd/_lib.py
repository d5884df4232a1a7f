"""
ctypes binding of the amtx C ABI (include/amtx.h -> amt_tools_amd/csrc/libamtx.so).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Pointers handed to the library are raw device addresses (`tensor.data_ptr()`), the stream is
the current torch HIP stream's handle.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AMTX_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libamtx.so')      # AMTX_LIB_PATH: a debug / timing build of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'amtx.h')

_lib = None


class AmtxError(RuntimeError):
    pass


ERR_UNSUPPORTED = -3       # AMTX_ERR_UNSUPPORTED (include/amtx.h)


def declared_symbols():
    """Every function name include/amtx.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(amtx_[a-z0-9_]+)\s*\(', text)))


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

_SIGNATURES = {
    'amtx_last_error': (C.c_char_p, []),
    'amtx_version': (_I, []),
    'amtx_spec_plan_create': (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, _I, _I]),
    'amtx_spec_plan_destroy': (_I, [_P]),
    'amtx_spec_num_bins': (_I, [_P]),
    'amtx_spec_num_frames': (_L, [_P, _L]),
    'amtx_spec_filterbank': (_I, [_P, _P]),
    'amtx_spec_power': (_I, [_P, _P, _L, _L, _I, _P, _P, _P]),
    'amtx_spec_scale': (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _P, _P]),
    'amtx_of_model_create': (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I]),
    'amtx_has_f16': (_I, []),
    'amtx_of_model_destroy': (_I, [_P]),
    'amtx_of_model_set_tensor': (_I, [_P, C.c_char_p, _P, _L]),
    'amtx_of_model_finalize': (_I, [_P]),
    'amtx_of_model_set_tensor_device': (_I, [_P, C.c_char_p, _P, _L]),
    'amtx_of_model_finalize_device': (_I, [_P, _P]),
    'amtx_of_workspace_bytes': (C.c_size_t, [_P, _I, _I]),
    'amtx_of_forward': (_I, [_P, _P, _L, _L, _L, _L, _I, _I, _P, C.c_size_t, _P, _P, _P, _P, _P, _P]),
    'amtx_of_fuses_db_scale': (_I, [_P]),
    'amtx_of_conv_stack_fused': (_I, [_P, _I, _I]),
    'amtx_of_forward_power': (_I, [_P, _P, _L, _L, _L, _P, _P, _I, _I, _P, C.c_size_t, _P, _P, _P, _P, _P, _P]),
    'amtx_of_takes_feats16': (_I, [_P]),
    'amtx_of_forward_feats16': (_I, [_P, _P, _I, _I, _P, C.c_size_t, _P, _P, _P, _P, _P, _P]),
    'amtx_of_offsets': (_I, [_P, _P, C.c_size_t, _I, _I, _P, _P, _P]),
    'amtx_of_num_stages': (_I, []),
    'amtx_of_stage_name': (C.c_char_p, [_I]),
    'amtx_of_profile_enable': (_I, [_P, _I]),
    'amtx_of_profile_read': (_I, [_P, C.POINTER(C.c_double), C.POINTER(_I)]),
    'amtx_linear_packed_elems': (_L, [_I, _I, _I]),
    'amtx_linear_pack': (_I, [_P, _I, _I, _I, _P]),
    'amtx_linear_fwd': (_I, [_P, _L, _I, _P, _I, _P, _P, _L, _I, _L, _I, _I, _P]),
    'amtx_split_planes': (_I, [_P, _L, _I, _P, _I, _L, _L, _P]),
    'amtx_linear_fwd_split': (_I, [_P, _L, _L, _P, _P, _P, _L, _I, _L, _L, _I, _I, _P]),
    'amtx_conv3x3_packed_elems': (_L, [_I, _I]),
    'amtx_conv3x3_pack': (_I, [_P, _P, _I, _I, _P]),
    'amtx_conv3x3_fwd': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    'amtx_conv3x3g_packed_elems': (_L, [_I, _I, _I]),
    'amtx_conv3x3g_pack': (_I, [_P, _P, _I, _I, _I, _P]),
    'amtx_conv3x3g_fwd': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    'amtx_bilstm_h_packed_elems': (_L, [_I, _I]),
    'amtx_bilstm_h_pack': (_I, [_P, _P, _I, _I, _P]),
    'amtx_bilstm_h_fwd': (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _P]),
    'amtx_conv1_fwd': (_I, [_P, _L, _L, _L, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'amtx_bilstm_packed_elems': (_L, [_I]),
    'amtx_bilstm_pack': (_I, [_P, _P, _I, _P]),
    'amtx_bilstm_fwd': (_I, [_P, _P, _I, _I, _P, _I, _I, _P]),
    'amtx_bilstm_pack_device': (_I, [_P, _P, _I, _P, _P, _P]),
    'amtx_bilstm_train_fwd': (_I, [_P, _P, _I, _P, _P, _I, _I, _P]),
    'amtx_bilstm_train_bwd': (_I, [_P, _P, _P, _I, _P, _I, _I, _P]),
    'amtx_cqt_plan_create': (_I, [C.POINTER(_P), _I, _I, C.c_double, _I, _I, C.c_double, C.POINTER(C.c_double), _I, _I, _I]),
    'amtx_cqt_plan_destroy': (_I, [_P]),
    'amtx_cqt_num_harmonics': (_I, [_P]),
    'amtx_cqt_num_frames': (_L, [_P, _L]),
    'amtx_cqt_workspace_bytes': (C.c_size_t, [_P, _I, _L]),
    'amtx_cqt_forward': (_I, [_P, _P, _L, _L, _I, _I, _P, C.c_size_t, _P, _P]),
    'amtx_cqt_forward16': (_I, [_P, _P, _L, _L, _I, _I, _P, C.c_size_t, _P, _P]),
    'amtx_cqt_forward16_split': (_I, [_P, _P, _L, _L, _I, _I, _P, C.c_size_t, _P, _L, _P]),
    'amtx_bilstm_h_pack_device': (_I, [_P, _P, _I, _I, _P, _P, _P]),
    'amtx_bilstm_h_train_fwd': (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _P]),
    'amtx_bilstm_h_train_bwd': (_I, [_P, _P, _P, _I, _I, _P, _I, _I, _I, _P]),
    'amtx_bn_train_workspace_bytes': (C.c_size_t, [_I]),
    'amtx_bn_relu_pool_train_fwd': (_I, [_P, _L, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'amtx_bn_relu_pool_train_bwd': (_I, [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'amtx_bce_logits_loss_workspace_bytes': (C.c_size_t, [_I, _I, _I]),
    'amtx_bce_logits_loss': (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    'amtx_rms_norm_workspace_bytes': (C.c_size_t, [_I, _L]),
    'amtx_rms_norm': (_I, [_P, _L, _L, _I, _P, _L, _P, C.c_size_t, _P]),
    'amtx_spec_mel_layout': (_I, [_I, _I, _I, _I, _P, _P, _P]),
    'amtx_notes_decode': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'amtx_notes_rows': (_I, [_P, _P, _I, _I, _I, _P, _L, _I, _P, _P, _L, _P, _P]),
    'amtx_pianoroll_fwd': (_I, [_P, _L, _I, _I, _I, _I, _F, _P, _P]),
    'amtx_matmul_workspace_bytes': (C.c_size_t, [_L, _L, _L]),
    'amtx_matmul_f32': (_I, [_P, _L, _I, _P, _L, _I, _P, _P, _L, _L, _L, _L, _P, C.c_size_t, _P]),
    'amtx_linear_train_fwd': (_I, [_P, _L, _P, _L, _P, _P, _L, _L, _I, _I, _P]),
    'amtx_linear_bwd_workspace_bytes': (C.c_size_t, [_L, _I, _I]),
    'amtx_linear_bwd': (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _P, C.c_size_t, _P]),
    'amtx_conv3x3_train_workspace_bytes': (C.c_size_t, [_L, _I, _I, _I]),
    'amtx_conv3x3_train_fwd': (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, C.c_size_t, _P]),
    'amtx_conv3x3_bwd': (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, C.c_size_t, _P]),
}


def lib():
    """Load libamtx.so (once).  Raises AmtxError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmtxError(f'{LIB_PATH} is missing: build the HIP extension first '
                            f'(python -m amt_tools_amd.build, or __graft_entry__.build()). '
                            f'There is no CPU fallback for the product path.')
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise AmtxError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise AmtxError(f'{LIB_PATH} does not export {name}') from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc is None or rc < 0:
        msg = lib().amtx_last_error()
        raise AmtxError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
    return rc


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ------------------------------------------------------------------------------------------------------------------------------
# kernel workspaces with guard bands (tests/test_gpu_guards.py)
# ------------------------------------------------------------------------------------------------------------------------------
GUARD_BYTES = 0        # > 0 (a multiple of 256): every workspace below gets this many pattern-filled bytes in front of and behind it
_GUARD_PATTERN = 0xA5
_GUARDED = {}          # data_ptr of a guarded workspace view -> (weak reference to its full allocation, guard bytes)


def alloc_workspace(nbytes, device):
    """uint8 tensor of `nbytes` for a kernel workspace.  With GUARD_BYTES set it is the middle of a larger allocation whose first and last
    GUARD_BYTES hold a pattern: several kernels issue masked stores to scratch lines and loads from clamped addresses, and a store that
    leaves its workspace must not go unnoticed (guards_intact)."""
    import torch
    nbytes = int(nbytes)
    g = int(GUARD_BYTES)
    if g <= 0:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    assert g % 256 == 0
    full = torch.empty(nbytes + 2 * g, dtype=torch.uint8, device=device)
    full[:g] = _GUARD_PATTERN
    full[g + nbytes:] = _GUARD_PATTERN
    ws = full[g:g + nbytes]                        # a view: keeps `full` alive, data_ptr() is 256-byte aligned like the allocation
    import weakref
    for k in [k for k, (r, _) in _GUARDED.items() if r() is None]:
        del _GUARDED[k]
    _GUARDED[ws.data_ptr()] = (weakref.ref(full), g)
    return ws


def guards_intact(ws):
    """True when the bands around a workspace from alloc_workspace still hold the pattern.  Only tensors alloc_workspace itself handed out
    with guard bands are checked (they are recorded there); anything else -- an unguarded workspace, a plain view of some other buffer --
    has no bands and answers True."""
    rec = _GUARDED.get(ws.data_ptr())
    base = getattr(ws, '_base', None)
    if rec is None or base is None or rec[0]() is not base:
        return True
    g = rec[1]
    return bool((base[:g] == _GUARD_PATTERN).all().item()) and bool((base[g + ws.numel():] == _GUARD_PATTERN).all().item())
