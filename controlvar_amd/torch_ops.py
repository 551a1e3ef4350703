"""``torch.ops.cvar.*`` - the PyTorch custom-op layer of the drop-in boundary.

Every op is a thin registration over one entry point of the C ABI (include/cvar.h, bound by ctypes in ``_lib`` / ``ops``):
  * dispatch key CUDA (= HIP on ROCm): the hand-written gfx950 kernel, launched on torch's CURRENT stream;
  * dispatch key CPU: raises ``RuntimeError`` - the HIP library is the only implementation, there is no eager / CPU fallback;
  * Meta (fake) kernels: output shapes / dtypes, so the ops compose with tracing and ``torch.library.opcheck``;
  * a non-zero ``cvar_status`` surfaces as ``RuntimeError`` (``_lib.CvarError``), wrong dtypes / layouts as ``TypeError`` / ``ValueError``
    (the reference's ATen ops raise in the same situations; nothing aborts across the ABI);
  * autograd formulas for the ops the reference differentiates through its operator slots (``linear``: F.linear / fused_mlp_func,
    ``attention``: flash_attn_func / slow_attn, ``ln_modulate``: dropout_add_layer_norm) - all backward math is again C-ABI kernels.

``controlvar_amd.slots`` adapts these ops to the exact signatures of the reference's module-global operator slots
(models/basic_var.py:15-29).  Importing this module registers the ops; it is imported by ``controlvar_amd.slots`` and on demand by
``controlvar_amd.register_torch_ops()``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch.library import Library, register_autograd, register_fake

from . import ops as K
from ._lib import ACT_GELU_TANH, ACT_NONE

__all__ = ['OPS']

_LIB = Library('cvar', 'DEF')
OPS: List[str] = []                      # names of all registered ops (tests walk this list)


def _cpu_stub(name):
    def impl(*a, **k):
        raise RuntimeError(f'cvar::{name}: the gfx950 HIP library is the only implementation of this op (no CPU / eager fallback) - '
                           'move the tensors to the GPU')
    return impl


def _define(name: str, schema: str, cuda_impl, fake=None):
    _LIB.define(name + schema)
    _LIB.impl(name, cuda_impl, 'CUDA')
    _LIB.impl(name, _cpu_stub(name), 'CPU')
    if fake is not None:
        register_fake('cvar::' + name)(fake)
    OPS.append(name)


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _check_operand(t: torch.Tensor, what: str):
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f'{what}: dtype {t.dtype} not supported (float32 = exact-f32 MFMA parity mode, bfloat16 = throughput mode)')


# ------------------------------------------------------------------------------------------------------------------- linear / gemm
def _linear(x, weight, bias=None, act=0, gate=None, gate_rows=1, residual=None, out_dtype=None):
    _check_operand(x, 'linear: x')
    if weight.dtype != x.dtype:
        raise TypeError('linear: x and weight must share a dtype')
    if weight.dim() != 2 or x.shape[-1] != weight.shape[1]:
        raise ValueError(f'linear: x (..., {x.shape[-1]}) does not match weight {tuple(weight.shape)}')
    N, Kd = weight.shape
    x2 = x.reshape(-1, Kd).contiguous()
    M = x2.shape[0]
    out = torch.empty(M, N, device=x.device, dtype=out_dtype or x.dtype)
    kw = {}
    if gate is not None:
        g = gate.reshape(-1, N).float().contiguous()
        kw.update(gate=g, ldg=N if g.shape[0] > 1 else 0, gate_rows=gate_rows)
    if residual is not None:
        kw.update(residual=residual.reshape(M, N).contiguous())
    K.gemm(x2, weight.contiguous(), out, M=M, N=N, K=Kd, bias=bias.float().contiguous() if bias is not None else None, act=act, **kw)
    return out.view(*x.shape[:-1], N)


def _linear_fake(x, weight, bias=None, act=0, gate=None, gate_rows=1, residual=None, out_dtype=None):
    return x.new_empty(*x.shape[:-1], weight.shape[0], dtype=out_dtype or x.dtype)


_define('linear', '(Tensor x, Tensor weight, Tensor? bias=None, int act=0, Tensor? gate=None, int gate_rows=1, Tensor? residual=None, '
                  'ScalarType? out_dtype=None) -> Tensor', _linear, _linear_fake)


def _linear_grads(x, weight, dy, want_bias):
    """dx, dW, db of y = x W^T + b through the same MFMA GEMM on transposed operands (train.TrainEngine's scheme)."""
    N, Kd = weight.shape
    x2 = x.reshape(-1, Kd).contiguous()
    dy2 = dy.reshape(-1, N).to(x.dtype).contiguous()
    M, Mp = x2.shape[0], _pad8(x2.shape[0])
    dev, T = x.device, x.dtype
    wt = torch.empty(Kd, N, device=dev, dtype=T)
    K.transpose(weight.contiguous(), wt, 1, N, Kd, Kd)
    dx = torch.empty(M, Kd, device=dev, dtype=T)
    K.gemm(dy2, wt, dx, M=M, N=Kd, K=N)
    dw = torch.empty(N, Kd, device=dev, dtype=torch.float32)
    db = torch.empty(N, device=dev, dtype=torch.float32) if want_bias else None
    if T == torch.bfloat16 and N % 128 == 0 and Kd % 128 == 0 and 2 * M * max(N, Kd) < 2 ** 31 - 1:
        K.gemm_tn(dy2, x2, dw, T=M, Nn=N, Kk=Kd, colsum=db)             # token-major operands read in place (cvar_gemm_tn): no transposed copies; db from the same pass
        return dx.view_as(x), dw, db
    ta = torch.zeros(N, Mp, device=dev, dtype=T)
    tb = torch.zeros(Kd, Mp, device=dev, dtype=T)
    K.transpose(dy2, ta, 1, M, N, N, ld_out=Mp)
    K.transpose(x2, tb, 1, M, Kd, Kd, ld_out=Mp)
    K.gemm(ta, tb, dw, M=N, N=Kd, K=Mp)
    if want_bias:
        K.rowsum(ta, Mp, db, N, M)
    return dx.view_as(x), dw, db


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act, gate, gate_rows, residual, out_dtype = inputs
    if gate is not None or residual is not None:
        ctx.unsupported = 'gate / residual epilogues are inference-side fusions; differentiate the unfused form'
        return
    ctx.unsupported = None
    ctx.act, ctx.has_bias = act, bias is not None
    ctx.save_for_backward(x, weight, bias)


def _linear_backward(ctx, dy):
    if ctx.unsupported:
        raise RuntimeError('cvar::linear backward: ' + ctx.unsupported)
    x, weight, bias = ctx.saved_tensors
    with torch.no_grad():
        dy = dy.contiguous()
        if ctx.act == ACT_GELU_TANH:           # recompute the pre-activation (one extra GEMM instead of a saved tensor)
            pre = _linear(x, weight, bias, ACT_NONE, None, 1, None, None)
            dy = dy.to(x.dtype).clone()
            K.gelu_bwd(pre.contiguous(), dy)
        dx, dw, db = _linear_grads(x, weight, dy, ctx.has_bias)
    return (dx, dw.to(weight.dtype), db.to(bias.dtype) if db is not None else None, None, None, None, None, None)


register_autograd('cvar::linear', _linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------------------------- conv 3x3
def _conv3x3(x, weight, bias, B, Hin, Win, stride=1, up=0, residual=None, out_dtype=None):
    """x: NHWC (B*Hin*Win, Cin) activations; weight: (Cout, 9*Cin) packed [ky][kx][ci] (models.VQVAE._pack layout)."""
    _check_operand(x, 'conv3x3: x')
    cout, k9 = weight.shape
    cin = k9 // 9
    Hout = Hin * 2 if up else (Hin // 2 if stride == 2 else Hin)
    Wout = Win * 2 if up else (Win // 2 if stride == 2 else Win)
    M = B * Hout * Wout
    out = torch.empty(M, cout, device=x.device, dtype=out_dtype or x.dtype)
    K.gemm(x.contiguous(), weight.contiguous(), out, M=M, N=cout, K=k9, bias=bias.float().contiguous() if bias is not None else None,
           residual=residual, conv=dict(Hin=Hin, Win=Win, Cin=cin, Hout=Hout, Wout=Wout, stride=stride, up=up))
    return out


def _conv3x3_fake(x, weight, bias, B, Hin, Win, stride=1, up=0, residual=None, out_dtype=None):
    Hout = Hin * 2 if up else (Hin // 2 if stride == 2 else Hin)
    Wout = Win * 2 if up else (Win // 2 if stride == 2 else Win)
    return x.new_empty(B * Hout * Wout, weight.shape[0], dtype=out_dtype or x.dtype)


_define('conv3x3', '(Tensor x, Tensor weight, Tensor? bias, int B, int Hin, int Win, int stride=1, int up=0, Tensor? residual=None, '
                   'ScalarType? out_dtype=None) -> Tensor', _conv3x3, _conv3x3_fake)


# ------------------------------------------------------------------------------------------------------------------- adaLN
def _rows2d(t: torch.Tensor, C: int) -> torch.Tensor:
    t = t.reshape(-1, C)
    return t if (t.dtype == torch.float32 and t.stride(1) == 1) else t.float().contiguous()


def _ln_modulate(x, scale, shift, rows_per, eps, out_dtype):
    C = x.shape[-1]
    x2 = x.reshape(-1, C).float().contiguous()
    sc, sh = _rows2d(scale, C), _rows2d(shift, C)
    if sc.stride(0) != sh.stride(0):
        sc, sh = sc.contiguous(), sh.contiguous()
    M = x2.shape[0]
    if sc.shape[0] * rows_per < M:
        raise ValueError(f'ln_modulate: {sc.shape[0]} modulation rows x rows_per {rows_per} < {M} tokens')
    out = torch.empty(M, C, device=x.device, dtype=out_dtype)
    from . import _lib
    _lib.check(_lib.load().cvar_ln_modulate(x2.data_ptr(), sc.data_ptr(), sh.data_ptr(), sc.stride(0) if sc.shape[0] > 1 else 0, rows_per,
                                           out.data_ptr(), K.dt(out), M, C, eps, K._stream()), 'cvar_ln_modulate')
    return out.view(*x.shape[:-1], C)


_define('ln_modulate', '(Tensor x, Tensor scale, Tensor shift, int rows_per, float eps, ScalarType out_dtype) -> Tensor', _ln_modulate,
        lambda x, scale, shift, rows_per, eps, out_dtype: x.new_empty(x.shape, dtype=out_dtype))


def _ln_modulate_bwd(x, dy, scale, rows_per, eps):
    """-> (dx, dscale (R, C), dshift (R, C)) of out = LN(x) * (1 + scale) + shift"""
    C = x.shape[-1]
    x2 = x.reshape(-1, C).float().contiguous()
    _check_operand(dy, 'ln_modulate_bwd: dy')
    dy2 = dy.reshape(-1, C).contiguous()
    sc = _rows2d(scale, C).contiguous()
    M, R = x2.shape[0], sc.shape[0]
    dev = x.device
    dx = torch.empty(M, C, device=dev, dtype=torch.float32)
    dada = torch.zeros(R, 2 * C, device=dev, dtype=torch.float32)
    ws = torch.empty(K.train_ws_floats(M, R, C) + 16, device=dev, dtype=torch.float32)
    K.ln_modulate_bwd(x2, dy2, sc, 0, C if R > 1 else 0, rows_per, None, dx, dada, 0, C, 2 * C, M, C, eps, ws)
    return dx.view_as(x), dada[:, :C].contiguous(), dada[:, C:].contiguous()


_define('ln_modulate_bwd', '(Tensor x, Tensor dy, Tensor scale, int rows_per, float eps) -> (Tensor, Tensor, Tensor)', _ln_modulate_bwd,
        lambda x, dy, scale, rows_per, eps: (x.new_empty(x.shape, dtype=torch.float32), scale.new_empty(scale.reshape(-1, x.shape[-1]).shape, dtype=torch.float32),
                                             scale.new_empty(scale.reshape(-1, x.shape[-1]).shape, dtype=torch.float32)))


def _ln_setup(ctx, inputs, output):
    x, scale, shift, rows_per, eps, out_dtype = inputs
    ctx.rows_per, ctx.eps = rows_per, eps
    ctx.shapes = (scale.shape, shift.shape, scale.dtype, shift.dtype, x.dtype)
    ctx.save_for_backward(x, scale)


def _ln_backward(ctx, dy):
    x, scale = ctx.saved_tensors
    with torch.no_grad():
        dyc = dy.contiguous()
        if dyc.dtype not in (torch.float32, torch.bfloat16):
            dyc = dyc.float()
        dx, dsc, dsh = torch.ops.cvar.ln_modulate_bwd(x, dyc, scale, ctx.rows_per, ctx.eps)
    ssh, hsh, sdt, hdt, xdt = ctx.shapes
    return dx.to(xdt), dsc.reshape(ssh).to(sdt), dsh.reshape(hsh).to(hdt), None, None, None


register_autograd('cvar::ln_modulate', _ln_backward, setup_context=_ln_setup)


def _silu_cast(x, out_dtype):
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    K.silu_cast(x.float().contiguous(), out)
    return out


_define('silu_cast', '(Tensor x, ScalarType out_dtype) -> Tensor', _silu_cast, lambda x, out_dtype: x.new_empty(x.shape, dtype=out_dtype))


def _gate_residual_(x, f, gate, gate_rows, rowscale=None):
    C = x.shape[-1]
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError('gate_residual_: x must be a contiguous float32 residual stream')
    _check_operand(f, 'gate_residual_: f')
    g = _rows2d(gate, C)
    K.gate_residual(x, f.reshape(-1, C).contiguous(), g, 0, g.stride(0) if g.shape[0] > 1 else 0, gate_rows,
                    rowscale.float().contiguous() if rowscale is not None else None, x.numel() // C, C)


_define('gate_residual_', '(Tensor(a!) x, Tensor f, Tensor gate, int gate_rows, Tensor? rowscale=None) -> ()', _gate_residual_,
        lambda x, f, gate, gate_rows, rowscale=None: None)


# ------------------------------------------------------------------------------------------------------------------- attention
def _holes(flat):
    return [(int(flat[2 * i]), int(flat[2 * i + 1])) for i in range(len(flat) // 2)] if flat else None


def _attention(qkv, H, q_off, l, scale, lvl_end, rowwise=False, holes=()):
    """qkv: (R, Lmax, 3*H*64) arena (q | k | v thirds); queries = rows [q_off, q_off + l) -> (out (R*l, H*64), lse (R, H, l) fp32).
    lvl_end / holes (flattened lo, hi pairs, one per level): the visibility tables of cvar_attention (include/cvar.h)."""
    _check_operand(qkv, 'attention: qkv')
    if qkv.dim() != 3 or qkv.shape[2] != 3 * H * 64 or not qkv.is_contiguous():
        raise ValueError(f'attention: arena must be contiguous (R, Lmax, 3*H*64); got {tuple(qkv.shape)} for H={H}')
    R, Lmax, _ = qkv.shape
    out = torch.empty(R * l, H * 64, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(R, H, l, device=qkv.device, dtype=torch.float32)
    K.attention(qkv, out, R, H, Lmax, q_off, l, scale, list(lvl_end) or None, rowwise=rowwise or qkv.dtype == torch.float32, lse=lse, holes=_holes(list(holes)))
    return out, lse


_define('attention', '(Tensor qkv, int H, int q_off, int l, float scale, int[] lvl_end, bool rowwise=False, int[] holes=[]) -> (Tensor, Tensor)', _attention,
        lambda qkv, H, q_off, l, scale, lvl_end, rowwise=False, holes=(): (qkv.new_empty(qkv.shape[0] * l, H * 64),
                                                                          qkv.new_empty(qkv.shape[0], H, l, dtype=torch.float32)))


def _attention_bwd(qkv, o, dout, lse, H, scale, lvl_end, rowwise=False, holes=()):
    """gradient of attention over the WHOLE arena (q_off = 0, l = Lmax: the teacher-forced form) -> dqkv with the arena layout"""
    R, Lmax, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(R * H * Lmax + 16, device=qkv.device, dtype=torch.float32)
    K.attention_bwd(qkv, o.contiguous(), dout.to(qkv.dtype).contiguous(), lse.contiguous(), dqkv, ws, R, H, Lmax, Lmax, scale, list(lvl_end) or None,
                    rowwise=rowwise or qkv.dtype == torch.float32, holes=_holes(list(holes)))
    return dqkv


_define('attention_bwd', '(Tensor qkv, Tensor o, Tensor dout, Tensor lse, int H, float scale, int[] lvl_end, bool rowwise=False, int[] holes=[]) -> Tensor',
        _attention_bwd, lambda qkv, o, dout, lse, H, scale, lvl_end, rowwise=False, holes=(): qkv.new_empty(qkv.shape))


def _attn_setup(ctx, inputs, output):
    qkv, H, q_off, l, scale, lvl_end, rowwise, holes = inputs
    ctx.args = (H, q_off, l, scale, tuple(lvl_end), rowwise, qkv.shape[1], tuple(holes))
    ctx.save_for_backward(qkv, output[0], output[1])
    ctx.mark_non_differentiable(output[1])


def _attn_backward(ctx, dout, dlse):
    H, q_off, l, scale, lvl_end, rowwise, Lmax, holes = ctx.args
    if q_off != 0 or l != Lmax:
        raise RuntimeError('cvar::attention backward needs the whole-sequence form (q_off = 0, l = Lmax); the KV-cached form is inference-only')
    qkv, o, lse = ctx.saved_tensors
    with torch.no_grad():
        dqkv = torch.ops.cvar.attention_bwd(qkv, o, dout.contiguous(), lse, H, scale, list(lvl_end), rowwise, list(holes))
    return dqkv, None, None, None, None, None, None, None


register_autograd('cvar::attention', _attn_backward, setup_context=_attn_setup)


def _attention_kv(kv, q, H, q_off, scale, lvl_end, rowwise=False, holes=()):
    """inference form: kv = (R, Lmax, 2*H*64) K/V arena (k | v halves), q = (R, l, H*64) queries of positions [q_off, q_off + l)
    -> out (R*l, H*64).  No autograd (the KV-cached form is inference-only)."""
    _check_operand(kv, 'attention_kv: kv')
    _check_operand(q, 'attention_kv: q')
    if kv.dim() != 3 or kv.shape[2] != 2 * H * 64 or not kv.is_contiguous():
        raise ValueError(f'attention_kv: arena must be contiguous (R, Lmax, 2*H*64); got {tuple(kv.shape)} for H={H}')
    R, Lmax, _ = kv.shape
    if q.dim() != 3 or q.shape[0] != R or q.shape[2] != H * 64 or not q.is_contiguous() or q.dtype != kv.dtype:
        raise ValueError(f'attention_kv: q must be contiguous (R, l, H*64) in the arena dtype; got {tuple(q.shape)} {q.dtype}')
    l = q.shape[1]
    out = torch.empty(R * l, H * 64, device=kv.device, dtype=kv.dtype)
    K.attention(kv, out, R, H, Lmax, q_off, l, scale, list(lvl_end) or None, rowwise=rowwise or kv.dtype == torch.float32, holes=_holes(list(holes)), q=q)
    return out


_define('attention_kv', '(Tensor kv, Tensor q, int H, int q_off, float scale, int[] lvl_end, bool rowwise=False, int[] holes=[]) -> Tensor', _attention_kv,
        lambda kv, q, H, q_off, scale, lvl_end, rowwise=False, holes=(): q.new_empty(q.shape[0] * q.shape[1], H * 64))


def _attention_kv_prescaled(kv, q, H, q_off, lvl_end, holes=()):
    """cvar_attention_prescaled (ABI 14): as attention_kv, but the query rows already carry scale * log2(e) (the QKV GEMM's split_alpha, or
    cos_qk_norm's factor) - the kernel the bf16 inference path runs; bf16 only"""
    _check_operand(kv, 'attention_kv_prescaled: kv')
    if kv.dtype != torch.bfloat16 or q.dtype != torch.bfloat16:
        raise TypeError('attention_kv_prescaled: bfloat16 only (the float32 parity mode uses attention_kv with rowwise=True)')
    if kv.dim() != 3 or kv.shape[2] != 2 * H * 64 or not kv.is_contiguous():
        raise ValueError(f'attention_kv_prescaled: arena must be contiguous (R, Lmax, 2*H*64); got {tuple(kv.shape)} for H={H}')
    R, Lmax, _ = kv.shape
    if q.dim() != 3 or q.shape[0] != R or q.shape[2] != H * 64 or not q.is_contiguous():
        raise ValueError(f'attention_kv_prescaled: q must be contiguous (R, l, H*64); got {tuple(q.shape)}')
    l = q.shape[1]
    out = torch.empty(R * l, H * 64, device=kv.device, dtype=kv.dtype)
    K.attention(kv, out, R, H, Lmax, q_off, l, 1.0, list(lvl_end) or None, holes=_holes(list(holes)), q=q, prescaled=True)
    return out


_define('attention_kv_prescaled', '(Tensor kv, Tensor q, int H, int q_off, int[] lvl_end, int[] holes=[]) -> Tensor', _attention_kv_prescaled,
        lambda kv, q, H, q_off, lvl_end, holes=(): q.new_empty(q.shape[0] * q.shape[1], H * 64))


def _cos_qk_norm_(qkv, H, q_off, l, scale_mul):
    R, Lmax, _ = qkv.shape
    K.cos_qk_norm(qkv, R, H, Lmax, q_off, l, scale_mul.float().contiguous())


_define('cos_qk_norm_', '(Tensor(a!) qkv, int H, int q_off, int l, Tensor scale_mul) -> ()', _cos_qk_norm_, lambda qkv, H, q_off, l, scale_mul: None)


# ------------------------------------------------------------------------------------------------------------------- sampler
def _cfg_sample(logits, B, nrep, coef, top_k, top_p, seed, stage, n_draw=1):
    """logits (nrep*B, l, V) fp32 -> ids (n_draw*B, l) int32 (control_var.py:295-307,501-505; helpers.py:6-19)"""
    if logits.dtype != torch.float32 or logits.dim() != 3 or logits.shape[0] != nrep * B:
        raise ValueError('cfg_sample: logits must be float32 (nrep*B, l, V)')
    _, l, V = logits.shape
    idx = torch.empty(n_draw * B, l, device=logits.device, dtype=torch.int32)
    K.cfg_sample(logits.contiguous(), B, nrep, l, V, list(coef), top_k, top_p, seed, stage, n_draw, idx)
    return idx


_define('cfg_sample', '(Tensor logits, int B, int nrep, float[] coef, int top_k, float top_p, int seed, int stage, int n_draw=1) -> Tensor', _cfg_sample,
        lambda logits, B, nrep, coef, top_k, top_p, seed, stage, n_draw=1: logits.new_empty(n_draw * B, logits.shape[1], dtype=torch.int32))


# ------------------------------------------------------------------------------------------------------------------- quantizer pyramid
def _ms_encode(f, codebook, phi_w, phi_b, phi_map, patch_nums, up, down):
    """f (B, Cvae, S, S) fp32 -> (ids (B, sum pn^2) int32, f_hat (B, Cvae, S, S))   (quant.py:184-215)"""
    B, Cv, S, _ = f.shape
    Ltot = sum(p * p for p in patch_nums)
    idx = torch.empty(B, Ltot, device=f.device, dtype=torch.int32)
    fh = torch.empty_like(f, dtype=torch.float32)
    K.ms_encode(f.float().contiguous(), codebook, codebook.shape[0], phi_w, phi_b, list(phi_map), list(patch_nums), up, down, idx, fh, None, B, S, Cv)
    return idx, fh


_define('ms_encode', '(Tensor f, Tensor codebook, Tensor phi_w, Tensor phi_b, int[] phi_map, int[] patch_nums, Tensor up, Tensor down) -> (Tensor, Tensor)',
        _ms_encode, lambda f, codebook, phi_w, phi_b, phi_map, patch_nums, up, down: (f.new_empty(f.shape[0], sum(p * p for p in patch_nums), dtype=torch.int32),
                                                                                   f.new_empty(f.shape, dtype=torch.float32)))


def _ms_next_input_(f_hat, idx, codebook, phi_w, phi_b, up, down, pn, pn_next, phi_k, up_off, down_off, want_tok=True):
    """one get_next_autoregressive_input step (quant.py:243-260) on f_hat (nb, nmaps, Cvae, S, S), in place; returns the next scale's
    tokens (nb, nmaps*pn_next^2, Cvae) or an empty tensor"""
    nb, nmaps, Cv, S, _ = f_hat.shape
    tok = torch.empty(nb, nmaps * pn_next * pn_next, Cv, device=f_hat.device, dtype=torch.float32) if want_tok else None
    K.ms_next_input(idx.to(torch.int32).contiguous(), codebook, phi_w, phi_b, up, down, f_hat, tok, nb, nmaps, pn, pn_next, S, Cv, phi_k, up_off, down_off)
    return tok if tok is not None else f_hat.new_empty(0)


_define('ms_next_input_', '(Tensor(a!) f_hat, Tensor idx, Tensor codebook, Tensor phi_w, Tensor phi_b, Tensor up, Tensor down, int pn, int pn_next, int phi_k, '
                          'int up_off, int down_off, bool want_tok=True) -> Tensor', _ms_next_input_,
        lambda f_hat, idx, codebook, phi_w, phi_b, up, down, pn, pn_next, phi_k, up_off, down_off, want_tok=True:
        f_hat.new_empty(f_hat.shape[0], f_hat.shape[1] * pn_next * pn_next, f_hat.shape[2]) if want_tok else f_hat.new_empty(0))


# ------------------------------------------------------------------------------------------------------------------- VQVAE glue
def _groupnorm_silu(x, weight, bias, B, HW, groups, eps, silu=True):
    """x: NHWC (B*HW, C) -> GroupNorm(groups, eps, affine) [+ SiLU]   (vae_modules.py:18-19,58-59)"""
    _check_operand(x, 'groupnorm_silu: x')
    C = x.shape[-1]
    ws = torch.empty(K.groupnorm_ws_bytes(B, HW, C), device=x.device, dtype=torch.uint8)
    out = torch.empty_like(x)
    K.groupnorm_silu(x.contiguous(), weight.float().contiguous(), bias.float().contiguous(), out, B, HW, C, groups, eps, silu, ws)
    return out


_define('groupnorm_silu', '(Tensor x, Tensor weight, Tensor bias, int B, int HW, int groups, float eps, bool silu=True) -> Tensor', _groupnorm_silu,
        lambda x, weight, bias, B, HW, groups, eps, silu=True: x.new_empty(x.shape))


# ------------------------------------------------------------------------------------------------------------------- training pieces
def _ce_fwd_bwd(logits, target, weight, gscale, grad_dtype):
    """CrossEntropyLoss(reduction='none') fused with its gradient -> (loss_tok (M,), dlogits (M, V))"""
    if logits.dtype != torch.float32:
        raise TypeError('ce_fwd_bwd: logits must be float32')
    V = logits.shape[-1]
    lg = logits.reshape(-1, V).contiguous()
    M = lg.shape[0]
    loss = torch.empty(M, device=lg.device, dtype=torch.float32)
    dl = torch.empty(M, V, device=lg.device, dtype=grad_dtype)
    K.ce_fwd_bwd(lg, target.reshape(-1).to(torch.int32).contiguous(), weight.reshape(-1).float().contiguous() if weight is not None else None, gscale, loss, dl, M, V)
    return loss, dl


_define('ce_fwd_bwd', '(Tensor logits, Tensor target, Tensor? weight, float gscale, ScalarType grad_dtype) -> (Tensor, Tensor)', _ce_fwd_bwd,
        lambda logits, target, weight, gscale, grad_dtype: (logits.new_empty(logits.numel() // logits.shape[-1], dtype=torch.float32),
                                                          logits.new_empty(logits.numel() // logits.shape[-1], logits.shape[-1], dtype=grad_dtype)))


def _adamw_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gscale=1.0):
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError('adamw_: parameters, gradients and moments are contiguous float32')
    K.adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, None, gscale)


_define('adamw_', '(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, float weight_decay, int step, '
                  'float gscale=1.0) -> ()', _adamw_, lambda p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, gscale=1.0: None)
