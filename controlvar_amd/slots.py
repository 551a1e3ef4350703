"""Drop-in implementations of the reference's L0 operator slots (models/basic_var.py:15-29).

The reference picks its fast operators by assigning module globals in ``models/basic_var.py``:

    dropout_add_layer_norm, fused_mlp_func     (flash_attn.ops)        basic_var.py:17-18, used at :44-49 and :163-171
    memory_efficient_attention                 (xformers.ops)          basic_var.py:20,   used at :114-115
    flash_attn_func                            (flash_attn)            basic_var.py:22,   used at :111-113   q, k, v: B L H c
    slow_attn                                  (F.scaled_dot_product_attention)  :24,    used at :117        q, k, v: B H L c

The functions below have exactly those names, argument names and layouts and run on the gfx950 kernels through
``torch.ops.cvar.*`` (controlvar_amd/torch_ops.py -> C ABI).  ``install(basic_var_module)`` assigns them into the reference's module so
that its own class tree (FFN / SelfAttention / SABlock) calls them; nothing in the reference has to be edited.  They are differentiable
where the reference differentiates through the slot (fused MLP, attention, fused add + LayerNorm).  float32 tensors take the exact-f32
path, bfloat16 the throughput path; float16 is not supported (the reference's HPU/GPU runs use bf16 autocast).
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch

from . import torch_ops  # noqa: F401  (registers torch.ops.cvar.*)
from ._lib import ACT_GELU_TANH, ACT_NONE

cvar = torch.ops.cvar

__all__ = ['fused_mlp_func', 'flash_attn_func', 'slow_attn', 'memory_efficient_attention', 'dropout_add_layer_norm', 'install']


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f'{what}: controlvar_amd operator slots run on the GPU only (no CPU fallback)')


# ------------------------------------------------------------------------------------------------------------------------ autocast
# The reference trains and validates under torch.autocast(bfloat16) with float32 Parameters (train_control_var_hpu.py:208).  ATen ops are
# cast by the autocast dispatcher; flash-attn's fused_mlp_func does it itself (custom_fwd: x, both weights and biases go to the autocast
# dtype).  torch.ops.cvar.* have no autocast rule, so the slot wrappers cast here: without it a bf16 activation meets a float32 weight
# (TypeError in cvar::linear), or - in the adaLN path, whose LayerNorm output is float32 - the FFN would silently take the exact-f32 GEMM.
_CAST_CACHE: dict = {}


def _autocast_dtype(t: torch.Tensor):
    dev = t.device.type
    return torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else None


def _base_of(t: torch.Tensor) -> torch.Tensor:
    return t._base if t._base is not None else t


def _cached(cache: dict, t: torch.Tensor, extra: tuple, make):
    """memo of make() per tensor VALUE identity: keyed by the python object of the tensor's base (a weak reference - the entry dies with the
    tensor, so a later tensor that is handed the same address can never inherit it), the view geometry and the version counter"""
    base = _base_of(t)
    key = (id(base), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype) + extra
    ent = cache.get(key)
    if ent is not None and ent[0]() is base and ent[1] == base._version and ent[2] == t.data_ptr():
        return ent[3]
    val = make()
    cache[key] = (weakref.ref(base, lambda _r, k=key, c=cache: c.pop(k, None)), base._version, t.data_ptr(), val)
    return val


def _cast(t: Optional[torch.Tensor], dtype):
    """t in `dtype`.  Differentiable for a parameter that needs a gradient; otherwise (inference) the cast copy is cached per tensor object
    and version the way the autocast dispatcher caches its weight casts, so a weight is converted once, not once per call."""
    if t is None or dtype is None or t.dtype == dtype or not t.is_floating_point():
        return t
    if t.requires_grad and torch.is_grad_enabled():
        return t.to(dtype)
    # memoise nn.Parameters only (tensors that outlive the call).  Activations are cast directly: a cached bf16 copy of
    # every x / q / k / v would stay alive until its source dies, and an inference tensor (torch.inference_mode) has no version
    # counter to key the cache on - reading ._version on one raises.
    if t.is_inference() or not isinstance(t, torch.nn.Parameter):
        return t.detach().to(dtype)
    return _cached(_CAST_CACHE, t, (dtype,), lambda: t.detach().to(dtype))


# --------------------------------------------------------------------------------------------------------------------- fused MLP
def fused_mlp_func(x, weight1, weight2, bias1=None, bias2=None, activation='gelu_approx', save_pre_act=True, return_residual=False,
                   checkpoint_lvl=0, heuristic=0, process_group=None):
    """flash_attn.ops.fused_dense.fused_mlp_func as FFN.forward calls it (basic_var.py:44-49): fc2(act(fc1(x))).
    GELU(tanh) runs in fc1's GEMM epilogue.  save_pre_act / checkpoint_lvl / heuristic only steer what upstream keeps for its
    backward; here the backward recomputes the pre-activation with one extra GEMM, so they are accepted and ignored."""
    _need_cuda(x, 'fused_mlp_func')
    if activation not in ('gelu_approx', 'relu'):
        raise ValueError(f'fused_mlp_func: activation {activation!r} (upstream accepts gelu_approx / relu; only gelu_approx is built)')
    if activation != 'gelu_approx':
        raise NotImplementedError('fused_mlp_func: only activation="gelu_approx" (the one FFN passes) is built')
    if process_group is not None:
        raise NotImplementedError('fused_mlp_func: tensor-parallel process groups are out of scope (replicas only, SURVEY.md 8e)')
    ac = _autocast_dtype(x)
    if ac is not None:                                   # flash-attn's custom_fwd: inputs AND weights in the autocast dtype
        x, weight1, weight2 = _cast(x, ac), _cast(weight1, ac), _cast(weight2, ac)
    h = cvar.linear(x, weight1, bias1, ACT_GELU_TANH)
    y = cvar.linear(h, weight2, bias2, ACT_NONE)
    return (y, x) if return_residual else y


# --------------------------------------------------------------------------------------------------------------------- attention
_LEVELS_CACHE: dict = {}


def _prefix_levels(attn_mask: torch.Tensor, Lq: int, Lk: int):
    """cached front of _prefix_levels_uncached: the reference passes the SAME registered buffer (attn_bias_for_masking, or a view of it) on
    every layer of every step, and decoding it costs a device->host copy (a sync) plus a Python walk over its rows - once per (buffer,
    version, view geometry) instead of once per layer per forward.  See _cached for why the key is the tensor object, not its address."""
    return _cached(_LEVELS_CACHE, attn_mask, (Lq, Lk), lambda: _prefix_levels_uncached(attn_mask, Lq, Lk))


def _prefix_levels_uncached(attn_mask: torch.Tensor, Lq: int, Lk: int):
    """The kernels implement 'query at position p sees keys [0, end(level(p))) minus one hole of its level', with level(p) = the first
    level whose end lies above p - exactly the structure of the reference's attn_bias_for_masking (control_var.py:158-191: plain,
    separate_decoding, separate_decoding + indep) and of its row slices.  Recover (level ends, holes) from an additive {0, -inf} mask
    and verify that it has that form; anything else is not something the reference passes and is refused, not approximated."""
    m = attn_mask
    while m.dim() > 2:
        if m.shape[0] != 1 and m.stride(0) != 0:         # an .expand()-ed broadcast (basic_var.py:115) is still one mask
            raise NotImplementedError('attention slot: per-batch / per-head masks are not built (the reference broadcasts one (1,1,L,L) mask)')
        m = m[0]
    if tuple(m.shape) != (Lq, Lk):
        raise ValueError(f'attention slot: mask shape {tuple(attn_mask.shape)} does not match (L_q={Lq}, L_k={Lk})')
    vis = (m == 0)
    if not bool((vis | torch.isneginf(m)).all()):
        raise NotImplementedError('attention slot: only additive {0, -inf} masks are built')
    v = vis.cpu().numpy()
    q_off = Lk - Lq
    rows = []
    for i in range(Lq):
        on = v[i].nonzero()[0]
        if len(on) == 0:
            raise NotImplementedError('attention slot: a query with no visible key')
        n = int(on[-1]) + 1
        off = (~v[i, :n]).nonzero()[0]
        if len(off) and (off[-1] - off[0] + 1 != len(off)):
            raise NotImplementedError('attention slot: more than one invisible run inside the visible prefix')
        rows.append((n, (int(off[0]), int(off[-1]) + 1) if len(off) else (0, 0)))
    ends = sorted(set(n for n, _ in rows))
    holes = {}
    for i, (n, hole) in enumerate(rows):
        pos = q_off + i
        want = next((e for e in ends if e > pos), None)
        if want != n:
            raise NotImplementedError('attention slot: mask is not block-causal over contiguous levels (control_var.py:158-191 form)')
        if holes.setdefault(n, hole) != hole:
            raise NotImplementedError('attention slot: queries of one level must share their hole')
    hl = [holes[e] for e in ends]
    return ends, (hl if any(b > a for a, b in hl) else None)


def _attention_blhc(q, k, v, scale: float, attn_mask=None, dropout_p: float = 0.0):
    """q (B, Lq, H, c), k / v (B, Lk, H, c) -> (B, Lq, H, c).  The queries are the LAST Lq positions of the key sequence (KV-cache
    convention of SelfAttention.forward, basic_var.py:106-108)."""
    _need_cuda(q, 'attention slot')
    if dropout_p:
        raise NotImplementedError('attention dropout is not built (attn_drop_rate = 0 in every shipped config, control_var.py:27)')
    B, Lq, H, c = q.shape
    Lk = k.shape[1]
    if c != 64:
        raise NotImplementedError('attention slot: head_dim must be 64 (embed_dim = 64 * depth in every reference model)')
    if k.shape != v.shape or k.shape[0] != B or k.shape[2] != H or Lq > Lk:
        raise ValueError(f'attention slot: inconsistent shapes q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)}')
    ac = _autocast_dtype(q)
    if ac is not None:                                   # F.scaled_dot_product_attention is on autocast's lower-precision list
        q, k, v = _cast(q, ac), _cast(k, ac), _cast(v, ac)
    if q.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f'attention slot: dtype {q.dtype} not supported (float32 or bfloat16)')
    q_off = Lk - Lq
    lvl_end, holes = _prefix_levels(attn_mask, Lq, Lk) if attn_mask is not None else ([], None)
    # pack the (q | k | v) arena the kernel reads: the model path writes this layout straight from the QKV GEMM; the slot pays a copy
    qfull = q if q_off == 0 else torch.cat((q.new_zeros(B, q_off, H, c), q), dim=1)
    arena = torch.stack((qfull, k.to(q.dtype), v.to(q.dtype)), dim=2).reshape(B, Lk, 3 * H * c).contiguous()
    out, _ = cvar.attention(arena, H, q_off, Lq, float(scale), lvl_end, False, [x for h in holes for x in h] if holes else [])
    return out.view(B, Lq, H, c)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **unused):
    """flash_attn.flash_attn_func as called at basic_var.py:113 - q, k, v: (B, L, H, c), returns (B, L, H, c)."""
    if causal:
        raise NotImplementedError('flash_attn_func: causal=True is never used by the reference (block-causal masks go through slow_attn)')
    scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
    return _attention_blhc(q, k, v, scale, None, dropout_p)


def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None, **unused):
    """xformers.ops.memory_efficient_attention as called at basic_var.py:115 - (B, L, H, c) layout, optional additive bias."""
    scale = scale if scale is not None else query.shape[-1] ** -0.5
    return _attention_blhc(query, key, value, scale, attn_bias, p)


def slow_attn(query, key, value, scale: Optional[float] = None, attn_mask=None, dropout_p=0.0, **unused):
    """F.scaled_dot_product_attention as called at basic_var.py:117 - q, k, v: (B, H, L, c), returns (B, H, L, c)."""
    scale = scale if scale is not None else query.shape[-1] ** -0.5
    o = _attention_blhc(query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2), scale, attn_mask, dropout_p)
    return o.transpose(1, 2)


# --------------------------------------------------------------------------------------------------------------------- fused add + LayerNorm
class _AddResidual(torch.autograd.Function):
    """residual_out = residual + x0 * rowscale * layerscale (fp32), in one kernel"""

    @staticmethod
    def forward(ctx, x0, residual, rowscale, layerscale):
        C = x0.shape[-1]
        rows = x0.numel() // C
        out = residual.float().clone().contiguous() if residual is not None else torch.zeros(x0.shape, device=x0.device, dtype=torch.float32)
        gate = layerscale.float().reshape(1, C) if layerscale is not None else torch.ones(1, C, device=x0.device)
        f = x0 if x0.dtype in (torch.float32, torch.bfloat16) else x0.float()
        if rowscale is not None:                     # per-token scale (drop_path over (B, L), basic_var.py:165,169): gate row = one token
            cvar.gate_residual_(out.view(rows, C), f.reshape(rows, C), gate, 1, rowscale.reshape(rows))
        else:
            cvar.gate_residual_(out.view(rows, C), f.reshape(rows, C), gate, rows, None)
        ctx.save_for_backward(x0, rowscale, layerscale)
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x0, rowscale, layerscale = ctx.saved_tensors
        C = x0.shape[-1]
        sc = g
        if layerscale is not None:
            sc = sc * layerscale.float()
        if rowscale is not None:
            sc = sc * rowscale.reshape(*x0.shape[:-1], 1).float()
        dls = None
        if layerscale is not None:
            t = g * x0.float()
            if rowscale is not None:
                t = t * rowscale.reshape(*x0.shape[:-1], 1).float()
            dls = t.reshape(-1, C).sum(0).to(layerscale.dtype)
        return sc.to(x0.dtype), (g if ctx.has_res else None), None, dls


def dropout_add_layer_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None, prenorm=False,
                           residual_in_fp32=False, return_dropout_mask=False):
    """flash_attn.ops.layer_norm.dropout_add_layer_norm as SABlock.fused_forward_wo_cond calls it (basic_var.py:163-171):
        residual_out = residual + drop(x0 * rowscale * layerscale);  y = LayerNorm(residual_out) * weight + bias
    returns y (dtype of x0), or (y, residual_out) with prenorm=True.  The affine LayerNorm runs on the adaLN kernel
    (LN(x) * (1 + (weight - 1)) + bias)."""
    _need_cuda(x0, 'dropout_add_layer_norm')
    if dropout_p:
        raise NotImplementedError('dropout_add_layer_norm: dropout_p > 0 is never used by the reference (basic_var.py:164: "no drop")')
    if return_dropout_mask:
        raise NotImplementedError('dropout_add_layer_norm: return_dropout_mask is not built')
    C = x0.shape[-1]
    res = _AddResidual.apply(x0, residual, rowscale, layerscale)
    out_dtype = x0.dtype if x0.dtype in (torch.float32, torch.bfloat16) else torch.float32
    rows = res.numel() // C
    y = cvar.ln_modulate(res, (weight.float() - 1.0).reshape(1, C), bias.float().reshape(1, C) if bias is not None else res.new_zeros(1, C),
                         rows, float(epsilon), out_dtype)
    if not residual_in_fp32 and residual is not None:
        res = res.to(residual.dtype)
    return (y, res) if prenorm else y


# --------------------------------------------------------------------------------------------------------------------- installation
def install(basic_var_module, flash: bool = True, fused: bool = True):
    """Assign the slots into the reference's ``models.basic_var`` module (or any module with the same globals).  Models built AFTER this
    call with flash_if_available / fused_if_available pick them up (FFN.__init__ :35, SelfAttention.__init__ :83-84, SABlock :141)."""
    if fused:
        basic_var_module.fused_mlp_func = fused_mlp_func
        basic_var_module.dropout_add_layer_norm = dropout_add_layer_norm
    if flash:
        basic_var_module.flash_attn_func = flash_attn_func
    basic_var_module.slow_attn = slow_attn
    return basic_var_module
