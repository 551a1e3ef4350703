"""Tensor-facing wrappers over the C ABI (include/cvar.h).

torch is plumbing here: it owns device memory and the stream; every computation is a call
into libcvar_hip.so with raw pointers.  All functions are asynchronous on the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ACT_GELU_TANH, ACT_NONE, CVAR_BF16, CVAR_F32, GemmDesc, check

_DT = {torch.float32: CVAR_F32, torch.bfloat16: CVAR_BF16}

# bench.py sets this to a list to collect (start_event, end_event, algorithmic_flops, shape tag) per GEMM launch
GEMM_PROFILE = None


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise TypeError(f'unsupported dtype {d}') from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.CvarError('controlvar_amd ops need device tensors (no CPU fallback)')
    return t.data_ptr()


_SPLITK_WS = {}
# per-call execution options of cvar_gemm (include/cvar.h): A/B measurement knobs, never needed for correctness
GEMM_TILE_CFG = 0          # 0 automatic, 1 128x128 only, 2 8-wave 256x256, 3 4-wave 256x256
GEMM_STAGGER = 0           # start-stagger window in shader cycles (0 = the library's default: off)
GEMM_GROUP_M = 0           # row tiles per scheduling group (0 = automatic)
SMALL_M_KERNEL = True      # False: small_m calls stay on the LDS-tiled kernels + split-K (A/B runs)


def ensure_splitk_workspace(device, nbytes: int = 256 << 20) -> torch.Tensor:
    """The split-K workspace ops.gemm hands to every cvar_gemm call on `device` issued from the current stream.  Caller-owned
    memory (the library keeps no state); one buffer per (device, stream), created on first use, so that GEMMs on different
    streams never share partial-sum storage and the split-K decision - hence the summation order - is the same on every stream
    (eager, side-stream warm-up, graph capture)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _SPLITK_WS[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    return ws


def splitk_workspace_keys() -> set:
    """(device index, stream handle) keys of the live split-K workspaces"""
    return set(_SPLITK_WS)


def take_splitk_workspace(key) -> Optional[torch.Tensor]:
    """remove one workspace from the per-stream table and hand it to the caller (models.graphed_generator: a workspace allocated during a
    graph capture belongs to that graph)"""
    return _SPLITK_WS.pop(key, None)


def release_splitk_workspace(device=None, stream_handle: Optional[int] = None):
    """Drop split-K workspaces (256 MB each) so the caching allocator can reuse the memory: the one of (device, stream_handle), or - with
    stream_handle None - every workspace of the device except the current stream's.  Streams come from a pool and their handles are
    reused; call this when a side stream is retired."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if stream_handle is not None:
        _SPLITK_WS.pop((idx, stream_handle), None)
        return
    cur = torch.cuda.current_stream(dev).cuda_stream
    for k in [k for k in _SPLITK_WS if k[0] == idx and k[1] != cur]:
        del _SPLITK_WS[k]


def gemm(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, lda: int = 0, ldw: int = 0, ldc: int = 0,
         bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0,
         gate: Optional[torch.Tensor] = None, ldg: int = 0, gate_rows: int = 1, gate_off: int = 0,
         residual: Optional[torch.Tensor] = None, ldr: int = 0,
         remap: Optional[Sequence[int]] = None, batch: int = 1, strideA: int = 0, strideW: int = 0, strideC: int = 0, strideR: int = 0,
         conv: Optional[dict] = None, a_off: int = 0, w_off: int = 0, c_off: int = 0,
         pre_act: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, gate_scale: Optional[torch.Tensor] = None,
         split: Optional[tuple] = None, split_alpha: float = 1.0, ln: Optional[tuple] = None, small_m: bool = False, split_k: bool = True,
         gn_part: Optional[torch.Tensor] = None):
    """C = epilogue(A @ W^T).  Offsets (*_off) are in elements of the respective tensor.
    gn_part (fp32 [B][tiles][N][3], conv only): GroupNorm partials of the output from the conv's epilogue (cvar_gemm_desc.gn_part, ABI 18; ask
    conv_gn_partials() whether the call can emit them - it fails loudly otherwise).
    ln = (out, ada, scale_off, shift_off, ld_ada, rows_per, eps): also write cast(LN(C[m]) * (1 + scale) + shift) of the finished rows to ``out`` -
    the ln_modulate of the op that follows (cvar_gemm_desc.ln_out, ABI 17; fused into the split-K reduction of small-M calls).
    split = (tensor, split_n, ld_split): result columns [0, split_n) go to ``tensor`` (rows not remapped), the rest to ``out`` at
    column n - split_n (cvar_gemm_desc.C_split; the qkv GEMM of inference: q beside a [R][Lmax][2C] K/V arena)."""
    d = GemmDesc()
    d.M, d.N, d.K, d.dtype = M, N, K, dt(A)
    if W.dtype != A.dtype:
        raise TypeError('A and W must share a dtype')
    d.A = _ptr(A) + a_off * A.element_size()
    d.W = _ptr(W) + w_off * W.element_size()
    d.lda, d.ldw = lda or K, ldw or K
    d.batch, d.strideA, d.strideW, d.strideC, d.strideR = batch, strideA, strideW, strideC, strideR
    if conv:
        d.conv = 1
        d.Hin, d.Win, d.Cin, d.Hout, d.Wout = conv['Hin'], conv['Win'], conv['Cin'], conv['Hout'], conv['Wout']
        d.stride, d.up = conv.get('stride', 1), conv.get('up', 0)
    d.alpha = alpha
    d.bias = _ptr(bias)
    d.act = act
    if gate is not None:
        d.gate = _ptr(gate) + gate_off * 4
        d.ldg, d.gate_rows = ldg, gate_rows
    if residual is not None:
        d.residual, d.res_dtype, d.ldr = _ptr(residual), dt(residual), ldr or N
    d.C = _ptr(out) + c_off * out.element_size()
    d.out_dtype, d.ldc = dt(out), ldc or N
    if remap is not None:
        d.remap_l, d.remap_L, d.remap_off = remap
    d.pre_act, d.aux, d.gate_scale = _ptr(pre_act), _ptr(aux), _ptr(gate_scale)
    if split is not None:
        st, d.split_n, d.ld_split = split
        if st.dtype != out.dtype:
            raise TypeError('split target and out must share a dtype')
        d.C_split = _ptr(st)
        d.split_alpha = float(split_alpha)
    if ln is not None:
        lo, ada, sc_off, sh_off, ld_ada, rows_per, eps = ln
        d.ln_out, d.ln_out_dtype = _ptr(lo), dt(lo)
        d.ln_scale, d.ln_shift = _ptr(ada) + 4 * sc_off, _ptr(ada) + 4 * sh_off
        d.ld_ln, d.ln_rows, d.ln_eps = ld_ada, rows_per, eps
    d.gn_part = _ptr(gn_part)
    ws = _SPLITK_WS.get((A.device.index, _stream()))
    if ws is None:
        ws = ensure_splitk_workspace(A.device)
    if split_k:
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    d.tile_cfg, d.stagger, d.group_m = (12 if (small_m and GEMM_TILE_CFG == 0 and SMALL_M_KERNEL) else GEMM_TILE_CFG), GEMM_STAGGER, GEMM_GROUP_M
    if GEMM_PROFILE is None:
        check(_lib.load().cvar_gemm(C.byref(d), _stream()), 'cvar_gemm')
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().cvar_gemm(C.byref(d), _stream()), 'cvar_gemm')
        e1.record()
        tag = ('conv' if conv else 'gemm', M, N, K, batch, 'gate' if gate is not None else ('res' if residual is not None else ('act' if act else ('remap' if remap is not None else 'plain'))), str(out.dtype)[6:])
        GEMM_PROFILE.append((e0, e1, 2.0 * M * N * K * batch, tag))
    return out


def gemm_tn(A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, T: int, Nn: int, Kk: int, lda: int = 0, ldb: int = 0, ldc: int = 0, c_off: int = 0,
            colsum: Optional[torch.Tensor] = None, colsum_off: int = 0):
    """out[n, k] (fp32) = sum_t A[t, n] * B[t, k] - the weight gradient dW = dY^T X on token-major bf16 operands (cvar_gemm_tn).
    colsum (fp32, optional): colsum[colsum_off + n] = sum_t A[t, n] - the bias gradient - from the same pass"""
    ws = _SPLITK_WS.get((A.device.index, _stream()))
    if ws is None:
        ws = ensure_splitk_workspace(A.device)
    cs = (_ptr(colsum) + 4 * colsum_off) if colsum is not None else None
    check(_lib.load().cvar_gemm_tn(_ptr(A), lda or Nn, _ptr(B), ldb or Kk, _ptr(out) + 4 * c_off, ldc or Kk, T, Nn, Kk, ws.data_ptr(), ws.numel(), cs, _stream()),
          'cvar_gemm_tn')
    return out


def ln_modulate(x: torch.Tensor, ada: torch.Tensor, scale_off: int, shift_off: int, ld_ada: int, rows_per: int,
                out: torch.Tensor, M: int, Cdim: int, eps: float):
    base = _ptr(ada)
    check(_lib.load().cvar_ln_modulate(_ptr(x), base + 4 * scale_off, base + 4 * shift_off, ld_ada, rows_per,
                                       _ptr(out), dt(out), M, Cdim, eps, _stream()), 'cvar_ln_modulate')
    return out


def silu_cast(x: torch.Tensor, out: torch.Tensor):
    check(_lib.load().cvar_silu_cast(_ptr(x), _ptr(out), dt(out), x.numel(), _stream()), 'cvar_silu_cast')
    return out


def _level_arrays(lvl_end, holes):
    """lvl_end: level ends; holes: optional [(lo, hi)] per level (keys the level's queries do not see) -> ctypes int arrays"""
    n = len(lvl_end) if lvl_end else 0
    arr = (C.c_int * max(n, 1))(*(lvl_end or [0]))
    harr = None
    if holes:
        if len(holes) != n:
            raise ValueError('one (lo, hi) hole per level')
        harr = (C.c_int * (2 * n))(*[int(v) for h in holes for v in h])
    return n, arr, harr


def attention(qkv: torch.Tensor, out: torch.Tensor, R: int, H: int, Lmax: int, q_off: int, l: int, scale: float,
              lvl_end: Optional[Sequence[int]] = None, qkv_off: int = 0, rowwise: bool = False, lse: Optional[torch.Tensor] = None,
              holes: Optional[Sequence[Sequence[int]]] = None, q: Optional[torch.Tensor] = None, prescaled: bool = False, v1: bool = False):
    """q=None: qkv is the packed arena [R][Lmax][3C]; q given: qkv is a K/V arena [R][Lmax][2C] and q holds the call's queries [R*l][C].
    prescaled: the query rows already carry scale * log2(e) (cvar_attention_prescaled; `scale` is then not used).  v1: the round-2 kernel."""
    n, arr, harr = _level_arrays(lvl_end, holes)
    if q is not None and (q.dtype != qkv.dtype or q.numel() < R * l * H * 64):
        raise ValueError('q must hold R*l rows of H*64 elements in the arena dtype')
    if prescaled:
        if rowwise or q is None:
            raise ValueError('prescaled queries: K/V-arena form of the MFMA kernel only')
        check(_lib.load().cvar_attention_prescaled(_ptr(qkv) + qkv_off * qkv.element_size(), _ptr(q), dt(qkv), R, H, Lmax, q_off, l,
                                                   arr, n, harr, _ptr(out), _ptr(lse), _stream()), 'cvar_attention_prescaled')
        return out
    fn = _lib.load().cvar_attention_rowwise if rowwise else (_lib.load().cvar_attention_v1 if v1 else _lib.load().cvar_attention)
    check(fn(_ptr(qkv) + qkv_off * qkv.element_size(), _ptr(q), dt(qkv), R, H, Lmax, q_off, l, scale,
                                     arr, n, harr, _ptr(out), _ptr(lse), _stream()), 'cvar_attention')
    return out


def attention_bwd(qkv, o, dout, lse, dqkv, ws, R, H, Lmax, l, scale, lvl_end=None, qkv_off: int = 0, rowwise: bool = False, holes=None):
    n, arr, harr = _level_arrays(lvl_end, holes)
    fn = _lib.load().cvar_attention_bwd_rowwise if rowwise else _lib.load().cvar_attention_bwd
    check(fn(_ptr(qkv) + qkv_off * qkv.element_size(), dt(qkv), _ptr(o), _ptr(dout), _ptr(lse), R, H, Lmax, 0, l, scale,
                                         arr, n, harr, _ptr(dqkv), _ptr(ws), _stream()), 'cvar_attention_bwd')
    return dqkv


def cos_qk_norm(qkv: torch.Tensor, R: int, H: int, Lmax: int, q_off: int, l: int, scale_mul: torch.Tensor,
                qkv_off: int = 0, sm_off: int = 0, norms: Optional[torch.Tensor] = None, q: Optional[torch.Tensor] = None, q_mul: float = 1.0):
    check(_lib.load().cvar_cos_qk_norm(_ptr(qkv) + qkv_off * qkv.element_size(), _ptr(q), dt(qkv), R, H, Lmax, q_off, l,
                                       _ptr(scale_mul) + 4 * sm_off, _ptr(norms), float(q_mul), _stream()), 'cvar_cos_qk_norm')


def cos_qk_norm_bwd(qkv, dqkv, R, H, Lmax, l, scale_mul, norms, dsm_tok, sm_off: int = 0):
    check(_lib.load().cvar_cos_qk_norm_bwd(_ptr(qkv), _ptr(dqkv), dt(qkv), R, H, Lmax, l, _ptr(scale_mul) + 4 * sm_off, _ptr(norms), _ptr(dsm_tok),
                                           _stream()), 'cvar_cos_qk_norm_bwd')


def cfg_sample(logits: torch.Tensor, B: int, nrep: int, l: int, V: int, coef: Sequence[float], top_k: int, top_p: float,
               seed: int, stage: int, n_draw: int, idx_out: torch.Tensor, combined: Optional[torch.Tensor] = None,
               margin: Optional[torch.Tensor] = None, kept: Optional[torch.Tensor] = None, seed_dev: Optional[torch.Tensor] = None,
               ldv: int = 0, codebook: Optional[torch.Tensor] = None, smooth_mul: float = 1.0, smooth_tau: float = 1.0,
               gumbel: Optional[torch.Tensor] = None, soft_out: Optional[torch.Tensor] = None):
    arr = (C.c_float * 4)(*(list(coef) + [0.0] * (4 - len(coef))))
    check(_lib.load().cvar_cfg_sample(_ptr(logits), B, nrep, l, V, arr, top_k, float(top_p), int(seed) & (2 ** 64 - 1), _ptr(seed_dev), stage, n_draw,
                                      _ptr(idx_out), _ptr(combined), _ptr(margin), _ptr(kept), int(ldv), _ptr(codebook),
                                      codebook.shape[1] if codebook is not None else 0, float(smooth_mul), float(smooth_tau), _ptr(gumbel), _ptr(soft_out),
                                      _stream()), 'cvar_cfg_sample')
    return idx_out


def ms_next_input(idx: torch.Tensor, codebook, phi_w, phi_b, up, down, f_hat, tok_out, nb, nmaps, pn, pn_next, S, Cvae,
                  phi_k: int, up_off: int, down_off: int):
    lib = _lib.load()
    pw = _ptr(phi_w) + 4 * phi_k * Cvae * 9 * Cvae
    pb = _ptr(phi_b) + 4 * phi_k * Cvae
    check(lib.cvar_ms_next_input(_ptr(idx), _ptr(codebook), pw, pb, _ptr(up) + 4 * up_off, _ptr(down) + 4 * down_off,
                                 _ptr(f_hat), _ptr(tok_out), nb, nmaps, pn, pn_next, S, Cvae, _stream()), 'cvar_ms_next_input')


def ms_encode(f, codebook, V, phi_w, phi_b, phi_map, patch_nums, up, down, idx_out, f_hat_out, margin_out, B, S, Cvae):
    n = len(patch_nums)
    pm = (C.c_int * n)(*phi_map)
    pn = (C.c_int * n)(*patch_nums)
    check(_lib.load().cvar_ms_encode(_ptr(f), _ptr(codebook), V, _ptr(phi_w), _ptr(phi_b), pm, pn, n, _ptr(up), _ptr(down),
                                     _ptr(idx_out), _ptr(f_hat_out), _ptr(margin_out), B, S, Cvae, _stream()), 'cvar_ms_encode')


def word_embed(tok, W, bias, lvl_pos, x, nb, nrep, l, Cvae, Cdim, x_rows, x_off, lvl_off: int = 0):
    check(_lib.load().cvar_word_embed(_ptr(tok), _ptr(W), _ptr(bias), _ptr(lvl_pos) + 4 * lvl_off * Cdim, _ptr(x), nb, nrep, l, Cvae, Cdim,
                                      x_rows, x_off, _stream()), 'cvar_word_embed')


def first_tokens(class_emb, cond_embed, labels, types, pos_start, lvl_pos, x, cond, R, first_l, Cdim, x_rows):
    check(_lib.load().cvar_first_tokens(_ptr(class_emb), _ptr(cond_embed), _ptr(labels), _ptr(types), _ptr(pos_start), _ptr(lvl_pos),
                                        _ptr(x), _ptr(cond), R, first_l, Cdim, x_rows, _stream()), 'cvar_first_tokens')


def groupnorm_ws_bytes(B, HW, Cdim) -> int:
    return int(_lib.load().cvar_groupnorm_ws_bytes(B, HW, Cdim))


def groupnorm_silu(x, weight, bias, out, B, HW, Cdim, groups, eps, silu, ws):
    check(_lib.load().cvar_groupnorm_silu(_ptr(x), dt(x), _ptr(weight), _ptr(bias), _ptr(out), B, HW, Cdim, groups, eps, int(silu),
                                          _ptr(ws), _stream()), 'cvar_groupnorm_silu')
    return out


def split3(x, out, M, Cdim, Cpad=0, ldx=0):
    """fp32 x[M][ldx] -> bf16 out[M][Cpad or 3 C] = [hi | lo | hi] (+ zero padding): the activation side of a split-bf16 product (cvar_split3, ABI 20)"""
    check(_lib.load().cvar_split3(_ptr(x), ldx or Cdim, _ptr(out), M, Cdim, Cpad or 3 * Cdim, _stream()), 'cvar_split3')
    return out


def groupnorm_silu_split3(x, weight, bias, out, B, HW, Cdim, groups, eps, silu, ws):
    check(_lib.load().cvar_groupnorm_silu_split3(_ptr(x), _ptr(weight), _ptr(bias), _ptr(out), B, HW, Cdim, groups, eps, int(silu), _ptr(ws), _stream()),
          'cvar_groupnorm_silu_split3')
    return out


def conv_gn_partials(dtype: torch.dtype, stride: int, Cin: int, Cout: int, Hin: int, Win: int, Hout: int, Wout: int):
    """(tiles_per_image, pixels_per_tile) when a 3x3 conv of this shape emits GroupNorm partials of its output (gemm(..., gn_part=...)), else None"""
    if GEMM_TILE_CFG not in (0, 6):
        return None
    nt, pp = C.c_int(0), C.c_int(0)
    if dtype not in _DT:
        return None
    ok = _lib.load().cvar_conv3x3_gn_partials(_DT[dtype], stride, Cin, Cout, Hin, Win, Hout, Wout, C.byref(nt), C.byref(pp))
    return (nt.value, pp.value) if ok else None


def groupnorm_silu_partials(x, weight, bias, out, B, HW, Cdim, groups, eps, silu, gn_part, tiles, tile_pixels, ws):
    check(_lib.load().cvar_groupnorm_silu_partials(_ptr(x), dt(x), _ptr(weight), _ptr(bias), _ptr(out), B, HW, Cdim, groups, eps, int(silu),
                                                   _ptr(gn_part), tiles, tile_pixels, _ptr(ws), _stream()), 'cvar_groupnorm_silu_partials')
    return out


def softmax_rows(s, p, rows, cols):
    check(_lib.load().cvar_softmax_rows(_ptr(s), _ptr(p), dt(p), rows, cols, _stream()), 'cvar_softmax_rows')
    return p


def transpose(inp, out, B, n, c, ld_in, in_off: int = 0, ld_out: int = 0):
    check(_lib.load().cvar_transpose(_ptr(inp) + in_off * inp.element_size(), _ptr(out), dt(inp), B, n, c, ld_in, ld_out or n, _stream()), 'cvar_transpose')
    return out


def nchw_to_nhwc(inp, out, B, Cdim, HW, Cpad):
    check(_lib.load().cvar_nchw_to_nhwc(_ptr(inp), _ptr(out), dt(out), B, Cdim, HW, Cpad, _stream()), 'cvar_nchw_to_nhwc')
    return out


def nhwc_to_nchw(inp, ld_in, out, B, Cdim, HW, lo=-3.0e38, hi=3.0e38, mul=1.0, add=0.0):
    check(_lib.load().cvar_nhwc_to_nchw(_ptr(inp), dt(inp), ld_in, _ptr(out), B, Cdim, HW, lo, hi, mul, add, _stream()), 'cvar_nhwc_to_nchw')
    return out


def embed_rows(idx: torch.Tensor, codebook: torch.Tensor, out: torch.Tensor):
    """out[n] = codebook[idx[n]] (int32 ids, fp32 rows)"""
    check(_lib.load().cvar_embed_rows(_ptr(idx), _ptr(codebook), codebook.shape[0], _ptr(out), idx.numel(), codebook.shape[1], _stream()), 'cvar_embed_rows')
    return out


def resample_sep(inp: torch.Tensor, wy: torch.Tensor, wx: torch.Tensor, out: torch.Tensor, B: int, h: int, w: int, H: int, W: int, Cdim: int):
    """NHWC fp32 separable resample with dense (H x h), (W x w) matrices"""
    check(_lib.load().cvar_resample_sep(_ptr(inp), _ptr(wy), _ptr(wx), _ptr(out), B, h, w, H, W, Cdim, _stream()), 'cvar_resample_sep')
    return out


# ------------------------------------------------------------------------------------------ training step
def gate_residual(x, f, gate, gate_off, ldg, gate_rows, rowscale, M, Cdim):
    check(_lib.load().cvar_gate_residual(_ptr(x), _ptr(f), dt(f), _ptr(gate) + 4 * gate_off, ldg, gate_rows, _ptr(rowscale), M, Cdim, _stream()), 'cvar_gate_residual')


def train_ws_floats(M: int, R: int, Cdim: int) -> int:
    """floats of workspace cvar_gated_grad / cvar_ln_modulate_bwd may use for M = R x l rows (cvar_train_ws_floats)"""
    return int(_lib.load().cvar_train_ws_floats(M, R, Cdim))


def gated_grad(dx, f, gate, gate_off, ldg, rowscale, df, dgate, dgate_off, ldo, R, l, Cdim, ws):
    check(_lib.load().cvar_gated_grad(_ptr(dx), _ptr(f), dt(f), _ptr(gate) + 4 * gate_off, ldg, _ptr(rowscale), _ptr(df), _ptr(dgate) + 4 * dgate_off, ldo,
                                      R, l, Cdim, _ptr(ws), ws.numel(), _stream()), 'cvar_gated_grad')


def gelu(a, h):
    check(_lib.load().cvar_gelu(_ptr(a), _ptr(h), dt(a), a.numel(), _stream()), 'cvar_gelu')
    return h


def gelu_bwd(a, dh):
    check(_lib.load().cvar_gelu_bwd(_ptr(a), _ptr(dh), dt(a), a.numel(), _stream()), 'cvar_gelu_bwd')
    return dh


def ln_modulate_bwd(x, dy, ada, scale_off, ld_ada, rows_per, dx_in, dx_out, dada, dscale_off, dshift_off, ldo, M, Cdim, eps, ws):
    check(_lib.load().cvar_ln_modulate_bwd(_ptr(x), _ptr(dy), dt(dy), _ptr(ada) + 4 * scale_off, ld_ada, rows_per, _ptr(dx_in), _ptr(dx_out),
                                           _ptr(dada) + 4 * dscale_off, _ptr(dada) + 4 * dshift_off, ldo, M, Cdim, eps, _ptr(ws), ws.numel(), _stream()), 'cvar_ln_modulate_bwd')


def colsum(A, lda, out, M, N, ws, accumulate=False, a_off: int = 0, out_off: int = 0):
    check(_lib.load().cvar_colsum(_ptr(A) + a_off * A.element_size(), dt(A), lda, _ptr(out) + 4 * out_off, M, N, int(accumulate), _ptr(ws), _stream()), 'cvar_colsum')


def wordembed_grad(dx, ldx: int, rows_per_sample: int, skip: int, tok, n_per_sample: int, B: int, Cdim: int, Cvae: int, out, w_off: int, b_off: int,
                   dx_off: int = 0):
    """dW (Cdim x Cvae at out[w_off:]) and db (out[b_off:]) of word_embed from the token-major fp32 tensors (cvar_wordembed_grad)"""
    lib = _lib.load()
    nb = lib.cvar_wordembed_grad_ws_bytes(B * n_per_sample, Cdim)
    ws = torch.empty(nb, device=dx.device, dtype=torch.uint8)
    check(lib.cvar_wordembed_grad(_ptr(dx) + 4 * dx_off, ldx, rows_per_sample, skip, _ptr(tok), n_per_sample, B, Cdim, Cvae,
                                  _ptr(out) + 4 * w_off, _ptr(out) + 4 * b_off, _ptr(ws), _stream()), 'cvar_wordembed_grad')


def rowsum(A, lda, out, nrows, ncols, accumulate=False, out_off: int = 0):
    check(_lib.load().cvar_rowsum(_ptr(A), dt(A), lda, _ptr(out) + 4 * out_off, nrows, ncols, int(accumulate), _stream()), 'cvar_rowsum')


def ce_fwd_bwd(logits, target, weight, gscale, loss_tok, dlogits, M, V):
    check(_lib.load().cvar_ce_fwd_bwd(_ptr(logits), _ptr(target), _ptr(weight), float(gscale), _ptr(loss_tok), _ptr(dlogits),
                                      dt(dlogits) if dlogits is not None else CVAR_F32, M, V, _stream()), 'cvar_ce_fwd_bwd')


def scatter_add_rows(src, ld_src, idx, dst, n, Cdim, src_off: int = 0):
    check(_lib.load().cvar_scatter_add_rows(_ptr(src) + 4 * src_off, ld_src, _ptr(idx), _ptr(dst), n, Cdim, _stream()), 'cvar_scatter_add_rows')


def silu_bwd(cond, dsilu, dcond):
    check(_lib.load().cvar_silu_bwd(_ptr(cond), _ptr(dsilu), _ptr(dcond), cond.numel(), _stream()), 'cvar_silu_bwd')


def adamw(p, g, m, v, lr, b1, b2, eps, wd, step, gscale_dev=None, gscale=1.0):
    check(_lib.load().cvar_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, b1, b2, eps, wd, step, _ptr(gscale_dev), gscale, _stream()), 'cvar_adamw')


def sumsq(x, partial, slot):
    check(_lib.load().cvar_sumsq(_ptr(x), x.numel(), _ptr(partial) + 8 * 256 * slot, _stream()), 'cvar_sumsq')


def adam_table(entries, device) -> torch.Tensor:
    """device table of cvar_adam_tensor {p, g, m, v, n, group, pad, w16} (7 x int64 per entry) for the multi-tensor optimizer calls;
    an entry is (p, g, m, v, group) or (p, g, m, v, group, w16) with w16 a contiguous bf16 tensor of p's size (or None)"""
    rows = []
    for (p, g, m, v, group, *rest) in entries:
        w16 = rest[0] if rest else None
        if w16 is not None and (w16.dtype != torch.bfloat16 or w16.numel() != p.numel() or not w16.is_contiguous()):
            raise ValueError('adam_table: the bf16 copy must be a contiguous bfloat16 tensor of the size of the parameter')
        rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), int(group) & 0xffffffff, w16.data_ptr() if w16 is not None else 0])
    return torch.tensor(rows, dtype=torch.int64).to(device)


def sumsq_multi(table: torch.Tensor, n: int, partial: torch.Tensor):
    check(_lib.load().cvar_sumsq_multi(_ptr(table), n, _ptr(partial), _stream()), 'cvar_sumsq_multi')


def adamw_multi(table: torch.Tensor, n: int, lrs: Sequence[float], wds: Sequence[float], b1, b2, eps, step, gscale_dev=None, gscale=1.0):
    la = (C.c_float * len(lrs))(*lrs)
    wa = (C.c_float * len(wds))(*wds)
    check(_lib.load().cvar_adamw_multi(_ptr(table), n, la, wa, len(lrs), b1, b2, eps, step, _ptr(gscale_dev), gscale, _stream()), 'cvar_adamw_multi')


def clip_coef(partial, count, pre_scale, max_norm, out2):
    check(_lib.load().cvar_clip_coef(_ptr(partial), count, pre_scale, max_norm, _ptr(out2), _stream()), 'cvar_clip_coef')


# ---- tokenizer input pipeline (preprocess.py)
def resample_u8(src: torch.Tensor, src_h: int, src_w: int, channels: int, axis: int, dst_extent: int, bounds: torch.Tensor,
                coeffs: torch.Tensor, ksize: int, dst: torch.Tensor):
    check(_lib.load().cvar_resample_u8(_ptr(src), src_h, src_w, channels, axis, dst_extent, _ptr(bounds), _ptr(coeffs), ksize, _ptr(dst),
                                       _stream()), 'cvar_resample_u8')
    return dst


def crop_flip_normalize(src: torch.Tensor, src_h: int, src_w: int, channels: int, top: int, left: int, out_h: int, out_w: int,
                        flip: bool, dst: torch.Tensor):
    check(_lib.load().cvar_crop_flip_normalize(_ptr(src), src_h, src_w, channels, top, left, out_h, out_w, int(flip), _ptr(dst),
                                               _stream()), 'cvar_crop_flip_normalize')
    return dst


def ignore_mask(cond: torch.Tensor, B: int, H: int, W: int, patch_nums: Sequence[int], first_masked_scale: int, image_first: int,
                out: torch.Tensor, L: int):
    arr = (C.c_int * len(patch_nums))(*patch_nums)
    check(_lib.load().cvar_ignore_mask(_ptr(cond), B, H, W, arr, len(patch_nums), first_masked_scale, image_first, _ptr(out), L,
                                       _stream()), 'cvar_ignore_mask')
    return out


def rle_paint(run_ends: torch.Tensor, ann_offsets: torch.Tensor, colours: torch.Tensor, n_ann: int, H: int, W: int, out: torch.Tensor):
    check(_lib.load().cvar_rle_paint(_ptr(run_ends), _ptr(ann_offsets), _ptr(colours), n_ann, H, W, _ptr(out), _stream()), 'cvar_rle_paint')
    return out
