"""ctypes binding of libcvar_hip.so (include/cvar.h).  No torch types cross this boundary:
callers hand over raw device pointers (``tensor.data_ptr()``), sizes and the HIP stream handle.

The library is mandatory on the product path: if it is missing or does not load, every op
raises (there is no CPU / eager fallback)."""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CVAR_LIB') or os.path.join(HERE, 'libcvar_hip.so')      # CVAR_LIB: A/B runs against another build
ABI_VERSION = 20

CVAR_F32, CVAR_BF16 = 0, 1
ACT_NONE, ACT_GELU_TANH, ACT_GELU_GRAD = 0, 1, 2

c_p = C.c_void_p
c_i = C.c_int
c_l = C.c_int64
c_f = C.c_float


class GemmDesc(C.Structure):
    """mirror of cvar_gemm_desc (include/cvar.h) - field order and types must match exactly"""
    _fields_ = [
        ('M', c_i), ('N', c_i), ('K', c_i), ('dtype', c_i),
        ('A', c_p), ('lda', c_l), ('W', c_p), ('ldw', c_l),
        ('batch', c_i),
        ('strideA', c_l), ('strideW', c_l), ('strideC', c_l), ('strideR', c_l),
        ('conv', c_i), ('Hin', c_i), ('Win', c_i), ('Cin', c_i), ('Hout', c_i), ('Wout', c_i), ('stride', c_i), ('up', c_i),
        ('alpha', c_f), ('bias', c_p), ('act', c_i),
        ('gate', c_p), ('ldg', c_l), ('gate_rows', c_i),
        ('residual', c_p), ('res_dtype', c_i), ('ldr', c_l),
        ('C', c_p), ('out_dtype', c_i), ('ldc', c_l),
        ('remap_l', c_i), ('remap_L', c_i), ('remap_off', c_i),
        ('pre_act', c_p), ('aux', c_p), ('gate_scale', c_p),
        ('ws', c_p), ('ws_bytes', c_l), ('tile_cfg', c_i), ('stagger', c_i), ('group_m', c_i),
        ('C_split', c_p), ('split_n', c_i), ('ld_split', c_l), ('split_alpha', c_f),
        ('ln_out', c_p), ('ln_out_dtype', c_i), ('ln_scale', c_p), ('ln_shift', c_p), ('ld_ln', c_l), ('ln_rows', c_i), ('ln_eps', c_f),
        ('gn_part', c_p),
    ]


# name -> (restype, argtypes); every symbol include/cvar.h declares
SIGNATURES = {
    'cvar_abi_version': (c_i, []),
    'cvar_status_str': (C.c_char_p, [c_i]),
    'cvar_gemm': (c_i, [C.POINTER(GemmDesc), c_p]),
    'cvar_resample_u8': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p]),
    'cvar_crop_flip_normalize': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'cvar_sumsq_multi': (c_i, [c_p, c_i, c_p, c_p]),
    'cvar_adamw_multi': (c_i, [c_p, c_i, C.POINTER(c_f), C.POINTER(c_f), c_i, c_f, c_f, c_f, c_i, c_p, c_f, c_p]),
    'cvar_rle_paint': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    'cvar_ignore_mask': (c_i, [c_p, c_i, c_i, c_i, C.POINTER(c_i), c_i, c_i, c_i, c_p, c_i, c_p]),
    'cvar_ln_modulate': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p, c_i, c_i, c_i, c_f, c_p]),
    'cvar_silu_cast': (c_i, [c_p, c_p, c_i, c_l, c_p]),
    'cvar_attention': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_attention_rowwise': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_attention_v1': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_attention_prescaled': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_attention_bwd': (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_attention_bwd_rowwise': (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(c_i), c_i, C.POINTER(c_i), c_p, c_p, c_p]),
    'cvar_cos_qk_norm': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_p]),
    'cvar_cos_qk_norm_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'cvar_cfg_sample': (c_i, [c_p, c_i, c_i, c_i, c_i, C.POINTER(c_f), c_i, c_f, C.c_uint64, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_p, c_p]),
    'cvar_ms_next_input': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cvar_ms_encode': (c_i, [c_p, c_p, c_i, c_p, c_p, C.POINTER(c_i), C.POINTER(c_i), c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'cvar_word_embed': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cvar_first_tokens': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cvar_groupnorm_ws_bytes': (c_l, [c_i, c_i, c_i]),
    'cvar_groupnorm_silu': (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p]),
    'cvar_groupnorm_silu_partials': (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_i, c_i, c_p, c_p]),
    'cvar_split3': (c_i, [c_p, c_l, c_p, c_l, c_i, c_i, c_p]),
    'cvar_groupnorm_silu_split3': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p]),
    'cvar_conv3x3_gn_partials': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(c_i), C.POINTER(c_i)]),
    'cvar_softmax_rows': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    'cvar_gemm_tn': (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_p, c_p]),
    'cvar_embed_rows': (c_i, [c_p, c_p, c_i, c_p, c_l, c_i, c_p]),
    'cvar_resample_sep': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cvar_transpose': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_l, c_p]),
    'cvar_nchw_to_nhwc': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cvar_nhwc_to_nchw': (c_i, [c_p, c_i, c_l, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_p]),
    # training step
    'cvar_gate_residual': (c_i, [c_p, c_p, c_i, c_p, c_l, c_i, c_p, c_l, c_i, c_p]),
    'cvar_train_ws_floats': (c_l, [c_l, c_i, c_i]),
    'cvar_gated_grad': (c_i, [c_p, c_p, c_i, c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_p]),
    'cvar_gelu': (c_i, [c_p, c_p, c_i, c_l, c_p]),
    'cvar_gelu_bwd': (c_i, [c_p, c_p, c_i, c_l, c_p]),
    'cvar_ln_modulate_bwd': (c_i, [c_p, c_p, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_p, c_l, c_p]),
    'cvar_colsum': (c_i, [c_p, c_i, c_l, c_p, c_l, c_i, c_i, c_p, c_p]),
    'cvar_wordembed_grad_ws_bytes': (c_l, [c_l, c_i]),
    'cvar_wordembed_grad': (c_i, [c_p, c_l, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'cvar_rowsum': (c_i, [c_p, c_i, c_l, c_p, c_i, c_i, c_i, c_p]),
    'cvar_ce_fwd_bwd': (c_i, [c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_l, c_i, c_p]),
    'cvar_scatter_add_rows': (c_i, [c_p, c_l, c_p, c_p, c_i, c_i, c_p]),
    'cvar_silu_bwd': (c_i, [c_p, c_p, c_p, c_l, c_p]),
    'cvar_adamw': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_p, c_f, c_p]),
    'cvar_sumsq': (c_i, [c_p, c_l, c_p, c_p]),
    'cvar_clip_coef': (c_i, [c_p, c_l, c_f, c_f, c_p, c_p]),
    # measurement aid (bench.py roofline.sustained_*)
    'cvar_probe_mfma_bf16': (c_i, [c_p, c_l, c_i, c_p, c_p]),
    'cvar_probe_mfma_bf16_32x32': (c_i, [c_p, c_l, c_i, c_p, c_p]),
    'cvar_probe_mfma_flops': (C.c_double, [c_i]),
}

_lib = None
_lock = threading.Lock()


class CvarError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and bind every symbol (raises if anything is missing)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(path):
            raise CvarError(f'{path} not found - build it with `python -m controlvar_amd.build` (hipcc, gfx950); '
                            'there is no fallback path')
        try:
            import torch  # noqa: F401  (loads the process-wide HIP runtime first so both sides share it)
        except Exception:  # pragma: no cover
            pass
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        v = lib.cvar_abi_version()
        if v != ABI_VERSION:
            raise CvarError(f'libcvar_hip.so ABI {v} != expected {ABI_VERSION}; rebuild')
        _lib = lib
        return lib


def check(status: int, what: str):
    if status != 0:
        msg = load().cvar_status_str(status).decode()
        raise CvarError(f'{what}: {msg} ({status})')
