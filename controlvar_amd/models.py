"""Host-side mirror of the reference's model API (models/__init__.py, models/vqvae.py,
models/var.py, models/control_var.py) over the HIP kernels of libcvar_hip.so.

Same class names, constructor keywords, method names / argument meaning and state_dict keys
as the reference, so a checkpoint (or a caller such as train_control_var_hpu.py:157-176,288,
322,332) works unchanged; the computation is organised MI355X-first instead of nn.Module-per-op:
weights are packed once into GEMM-ready device buffers, the KV cache is a preallocated arena
written by the QKV GEMM's epilogue, adaLN parameters are computed once per generation, and every
scale step runs as a short sequence of fused kernels.  There is no CPU / eager fallback.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import CvarError
from ._lib import ACT_GELU_TANH
from .pyramid import packed_tables
from .spec import DEFAULT_PATCH_NUMS, VaeConfig, VarConfig, attention_levels, vae_state_shapes, var_state_shapes
from .synth import synth_vae_state, synth_var_state


class _Tree(nn.Module):
    """plain container so that dotted state_dict keys map onto nested attributes"""


def _register_tree(root: nn.Module, shapes, values: Dict[str, torch.Tensor], requires_grad: bool):
    for key, (shape, kind) in shapes.items():
        parts = key.split('.')
        m = root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Tree())
            m = m._modules[p]
        t = values[key]
        assert tuple(t.shape) == tuple(shape), key
        if kind == 'param':
            m.register_parameter(parts[-1], nn.Parameter(t, requires_grad=requires_grad and t.is_floating_point()))
        else:
            m.register_buffer(parts[-1], t)


def _tensor_sig(module: nn.Module):
    """identity + in-place-version stamp of every parameter and buffer: changes when a tensor is replaced (load_state_dict(assign),
    .data swaps, .to()) or written in place through autograd-visible ops (torch.optim steps, p.mul_(), p.copy_())"""
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters(recurse=True)) + list(module.buffers(recurse=True)))


def _check_index_range(t: torch.Tensor, lo: int, hi: int, what: str):
    """nn.Embedding raises on an out-of-range index (the reference's failure mode); the gather kernels read unchecked, so ids are
    validated here: for free on host tensors, with one tiny reduction + read-back on device tensors (outside any graph capture)."""
    if t.numel() == 0:
        return
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    mn, mx = (int(v) for v in torch.stack((t.min(), t.max())).tolist())
    if mn < lo or mx > hi:
        raise IndexError(f'{what}: index out of range (got [{mn}, {mx}], valid [{lo}, {hi}])')


def _compute_dtype(d) -> torch.dtype:
    if isinstance(d, torch.dtype):
        return d
    return {'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16, 'fp32': torch.float32, 'f32': torch.float32, 'float32': torch.float32}[d]


# =====================================================================================
# VQVAE
# =====================================================================================
class VQVAE(nn.Module):
    """Multi-scale VQVAE tokenizer (reference: models/vqvae.py:16-109).

    compute_dtype: torch.bfloat16 (throughput mode: bf16 NHWC activations, fp32 accumulate) or
    torch.float32 (parity mode: exact-f32 MFMA).  The quantizer itself always runs in fp32.
    """

    def __init__(self, vocab_size=4096, z_channels=32, ch=128, dropout=0.0, beta=0.25, using_znorm=False, quant_conv_ks=3,
                 quant_resi=0.5, share_quant_resi=4, default_qresi_counts=0, v_patch_nums=DEFAULT_PATCH_NUMS, test_mode=True,
                 compute_dtype=torch.bfloat16, init_seed: int = 0, decode_chunk: int = 128, encoder_precision: Optional[str] = None):
        """encoder_precision (bf16 compute only; None = 'bf16'): arithmetic of the ENCODER conv stack in front of the exact quantizer -
        'bf16' (throughput), 'bf16x3' (split-bf16: every operand as hi + lo bf16, three MFMA products per multiply, fp32 accumulate and fp32 activations:
        ~2^-16 relative error per product, ids agree with the reference's fp32 encoder far beyond plain bf16 at about a third of its encoder rate) or 'fp32'
        (the parity mode's exact-f32 MFMA for the encoder only).  The decoder keeps compute_dtype either way."""
        super().__init__()
        if using_znorm or quant_conv_ks != 3 or abs(quant_resi - 0.5) > 1e-9 or share_quant_resi != 4 or dropout != 0.0:
            raise NotImplementedError('only the shipped VQVAE configuration (vqvae.py:18-27 defaults, share_quant_resi=4) is built')
        self.cfg = VaeConfig(vocab=vocab_size, z_channels=z_channels, ch=ch, share_quant_resi=share_quant_resi,
                             patch_nums=tuple(v_patch_nums))
        self.test_mode = test_mode
        self.V = self.vocab_size = vocab_size
        self.Cvae = z_channels
        self.downsample = 2 ** (len(self.cfg.ch_mult) - 1)
        self.compute_dtype = _compute_dtype(compute_dtype)
        ep = encoder_precision or ('bf16' if self.compute_dtype == torch.bfloat16 else 'fp32')
        if ep not in ('bf16', 'bf16x3', 'fp32') or (self.compute_dtype == torch.float32 and ep != 'fp32'):
            raise ValueError(f'encoder_precision={encoder_precision!r}: one of bf16 / bf16x3 / fp32 (an fp32 model encodes in fp32)')
        self.encoder_precision = ep
        # images per decoder pass (an image's bits do not depend on it).  128 since round 5: VQVAE round trip 1 355 -> 1 373-1 379 images/s, headline +0.4 % at the
        # same 226 GB peak (64 / 128 / 256: 201.5 / 202.3 / 202.0 images/s on one box)
        self.decode_chunk = decode_chunk
        _register_tree(self, vae_state_shapes(self.cfg), synth_vae_state(self.cfg, init_seed), requires_grad=not test_mode)
        self._packed = None
        if test_mode:
            self.eval()

    # ---- nn.Module plumbing
    def load_state_dict(self, state_dict: Dict[str, Any], strict=True, assign=False):
        key = 'quantize.ema_vocab_hit_SV'
        if key in state_dict and state_dict[key].shape[0] != self.quantize.ema_vocab_hit_SV.shape[0]:
            state_dict = dict(state_dict)
            state_dict[key] = self.quantize.ema_vocab_hit_SV          # vqvae.py:106-108
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.quant_conv.weight.device

    # ---- weight packing (one-time layout work, not on the timed path)
    def _state_sig(self):
        return _tensor_sig(self)

    def _pack(self, check: bool = False):
        """GEMM-ready device copies of the weights.  check=True (public entry points): rebuild them when any parameter / buffer
        was replaced or modified in place since they were made (optimizer steps, manual edits, .data swaps)."""
        if self._packed is not None and (not check or self._packed_sig == self._state_sig()):
            return self._packed
        dev, T = self.device, self.compute_dtype
        if dev.type != 'cuda':
            raise RuntimeError('controlvar_amd.VQVAE computes on the GPU only; call .to("cuda") first')
        kch = 8 if T == torch.bfloat16 else 4
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        P: Dict[str, Any] = {'conv': {}, 'norm': {}}
        for k, v in sd.items():
            if k.endswith('.weight') and v.dim() == 4 and not k.startswith('quantize.'):
                name = k[:-len('.weight')]
                fp32_only = name in ('quant_conv', 'post_quant_conv')        # quantizer side stays fp32
                Tw = torch.float32 if fp32_only else T
                kc = 4 if fp32_only else kch
                cout, cin, ks, _ = v.shape
                if ks == 3:
                    cin_p = (cin + kc - 1) // kc * kc
                    w = torch.zeros(cout, 3, 3, cin_p, device=dev, dtype=torch.float32)
                    w[..., :cin] = v.permute(0, 2, 3, 1)
                    w = w.reshape(cout, 9 * cin_p)
                else:
                    cin_p = cin
                    w = v.reshape(cout, cin)
                P['conv'][name] = dict(w=w.to(Tw).contiguous(), b=sd[name + '.bias'].float().contiguous(), cin=cin_p, cout=cout, ks=ks)
                if name.startswith('encoder.') and T == torch.bfloat16 and self.encoder_precision != 'bf16':
                    P.setdefault('enc', {})[name] = self._pack_encoder_conv(v, sd[name + '.bias'], dev)
            elif '.norm' in k and k.endswith('.weight'):
                name = k[:-len('.weight')]
                P['norm'][name] = (v.float().contiguous(), sd[name + '.bias'].float().contiguous())
        P['E'] = sd['quantize.embedding.weight'].float().contiguous()
        nphi = self.cfg.share_quant_resi
        pw = torch.stack([sd[f'quantize.quant_resi.qresi_ls.{k}.weight'] for k in range(nphi)])      # (k, co, ci, 3, 3)
        P['phi_w'] = pw.permute(0, 2, 3, 4, 1).reshape(nphi, self.Cvae, 9, self.Cvae).float().contiguous()   # [k][ci][tap][co]
        P['phi_b'] = torch.stack([sd[f'quantize.quant_resi.qresi_ls.{k}.bias'] for k in range(nphi)]).float().contiguous()
        # Phi as ONE conv for the generic conv path (low-resolution reconstructions): h (1 - r) + (conv(h) + b) r = conv'(h) + b r with the
        # identity folded into the centre tap, W' = r W + (1 - r) I  (quant.py:263-270)
        r = float(self.cfg.quant_resi)
        pf = pw.float() * r                                                                       # (k, co, ci, 3, 3)
        pf[:, torch.arange(self.Cvae), torch.arange(self.Cvae), 1, 1] += 1.0 - r
        P['phi_fold_w'] = pf.permute(0, 1, 3, 4, 2).reshape(nphi, self.Cvae, 9 * self.Cvae).contiguous()   # [k][co][tap][ci]
        P['phi_fold_b'] = (P['phi_b'] * r).contiguous()
        up, down, offs = packed_tables(self.cfg.patch_nums)
        P['up'] = torch.from_numpy(up).to(dev)
        P['down'] = torch.from_numpy(down).to(dev)
        P['tab_off'] = [int(o) for o in offs]
        P['phi_map'] = self.cfg.phi_map
        self._packed = P
        self._packed_sig = self._state_sig()
        return P

    def _pack_encoder_conv(self, v, bias, dev):
        """encoder weights of the non-default encoder precisions: 'fp32' = the parity mode's packing; 'bf16x3' = per tap [w_hi | w_hi | w_lo] over 3 x cin channels
        (the weight side of the split product, include/cvar.h cvar_split3)"""
        cout, cin, ks, _ = v.shape
        cin_p = (cin + 3) // 4 * 4
        w = torch.zeros(cout, ks, ks, cin_p, device=dev, dtype=torch.float32)
        w[..., :cin] = v.permute(0, 2, 3, 1).float()
        b = bias.float().contiguous()
        if self.encoder_precision == 'fp32':
            return dict(w=w.reshape(cout, ks * ks * cin_p).contiguous(), b=b, cin=cin_p, cout=cout, ks=ks)
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        cin3 = (3 * cin_p + 7) // 8 * 8                                       # bf16 K moves 16-byte chunks
        w3 = torch.zeros(cout, ks, ks, cin3, device=dev, dtype=torch.bfloat16)
        w3[..., :cin_p], w3[..., cin_p:2 * cin_p], w3[..., 2 * cin_p:3 * cin_p] = hi, hi, lo
        return dict(w=w3.reshape(cout, ks * ks * cin3).contiguous(), b=b, cin=cin_p, cin3=cin3, cout=cout, ks=ks)

    # ---- conv-stack building blocks (NHWC activations of compute dtype)
    # GroupNorm statistics from the producing conv's epilogue (round 5): a 3x3 conv that the LDS-halo kernel takes writes per-tile partial sums of its
    # output next to it; the GroupNorm that reads that tensor then skips its statistics pass (7 % of a VQVAE round trip).  The partials ride on the
    # tensor object (``_gn_part``): whoever consumes the tensor without a GroupNorm simply ignores them.  Class-level switch for A/B runs and tests.
    GN_FROM_CONV = True

    def _conv(self, x, name, B, Hin, Win, *, stride=1, up=0, residual=None, out_dtype=None):
        c = self._pack()['conv'][name]
        T = c['w'].dtype
        if c['ks'] == 1:
            M = B * Hin * Win
            out = torch.empty(M, c['cout'], device=x.device, dtype=out_dtype or T)
            # never split along K: with the tile kernels alone a row's sum order does not depend on M, so an image decodes / encodes to the same bits in any batch
            ops.gemm(x, c['w'], out, M=M, N=c['cout'], K=c['cin'], bias=c['b'], residual=residual, split_k=False)
            return out, Hin, Win
        Hout = Hin * 2 if up else (Hin // 2 if stride == 2 else Hin)
        Wout = Win * 2 if up else (Win // 2 if stride == 2 else Win)
        M = B * Hout * Wout
        out = torch.empty(M, c['cout'], device=x.device, dtype=out_dtype or T)
        geo = ops.conv_gn_partials(T, stride, c['cin'], c['cout'], Hin, Win, Hout, Wout) if (self.GN_FROM_CONV and out.dtype == T) else None
        part = torch.empty(B, geo[0], c['cout'], 3, device=x.device, dtype=torch.float32) if geo else None
        cv = dict(Hin=Hin, Win=Win, Cin=c['cin'], Hout=Hout, Wout=Wout, stride=stride, up=up)
        try:
            ops.gemm(x, c['w'], out, M=M, N=c['cout'], K=9 * c['cin'], bias=c['b'], residual=residual, conv=cv, gn_part=part)
        except CvarError as e:
            # conv_gn_partials() decides by shape alone; the kernel that emits the partials also needs 16-byte aligned bias / output / residual and dense
            # strides (a state dict assigned from a flat buffer can hand over a 4-byte aligned bias).  Such a call is unsupported WITH partials, not
            # without: run it on the implicit-GEMM tiles and let the GroupNorm that follows make its own statistics pass.
            if part is None or 'unsupported' not in str(e).lower():
                raise
            geo, part = None, None
            ops.gemm(x, c['w'], out, M=M, N=c['cout'], K=9 * c['cin'], bias=c['b'], residual=residual, conv=cv)
        if geo:
            out._gn_part = (part, geo[0], geo[1])
        return out, Hout, Wout

    def _gn(self, x, name, B, HW, C, silu=True):
        w, b = self._pack()['norm'][name]
        ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=x.device, dtype=torch.uint8)
        out = torch.empty_like(x)
        gp = getattr(x, '_gn_part', None)
        if gp is not None and gp[0].shape[0] == B and gp[0].shape[2] == C and gp[1] * gp[2] == HW:
            return ops.groupnorm_silu_partials(x, w, b, out, B, HW, C, self.cfg.gn_groups, self.cfg.gn_eps, silu, gp[0], gp[1], gp[2], ws)
        return ops.groupnorm_silu(x, w, b, out, B, HW, C, self.cfg.gn_groups, self.cfg.gn_eps, silu, ws)

    def _resblock(self, x, name, B, H, W, cin, cout):
        h = self._gn(x, name + '.norm1', B, H * W, cin)
        h, _, _ = self._conv(h, name + '.conv1', B, H, W)
        h = self._gn(h, name + '.norm2', B, H * W, cout)
        if cin != cout:
            x, _, _ = self._conv(x, name + '.nin_shortcut', B, H, W)
        h, _, _ = self._conv(h, name + '.conv2', B, H, W, residual=x)
        return h

    def _attnblock(self, x, name, B, HW, C):
        T = x.dtype
        n = self._gn(x, name + '.norm', B, HW, C, silu=False)
        qkv, _, _ = self._conv(n, name + '.qkv', B, HW, 1)                     # (B*HW, 3C): q | k | v
        kch = 16 // x.element_size()
        HWp = -(-HW // kch) * kch               # the GEMM's K dimension moves 16-byte chunks: pad the key axis of P and V^T with zeros
        vT = torch.empty(B, C, HW, device=x.device, dtype=T) if HWp == HW else torch.zeros(B, C, HWp, device=x.device, dtype=T)
        ops.transpose(qkv, vT, B, HW, C, 3 * C, in_off=2 * C, ld_out=HWp)
        s = torch.empty(B, HW, HW, device=x.device, dtype=torch.float32)
        ops.gemm(qkv, qkv, s, M=HW, N=HW, K=C, lda=3 * C, ldw=3 * C, w_off=C, alpha=float(int(C) ** -0.5), batch=B,
                 strideA=HW * 3 * C, strideW=HW * 3 * C, strideC=HW * HW)
        p = torch.empty(B, HW, HW, device=x.device, dtype=T)
        ops.softmax_rows(s, p, B * HW, HW)
        if HWp != HW:                           # latent sizes other than 16x16 (low-resolution reconstructions, vqvae.py:97-104 same_shape=False)
            pp = torch.zeros(B, HW, HWp, device=x.device, dtype=T)
            pp[:, :, :HW] = p
            p = pp
        o = torch.empty(B * HW, C, device=x.device, dtype=T)
        ops.gemm(p, vT, o, M=HW, N=C, K=HWp, batch=B, strideA=HW * HWp, strideW=C * HWp, strideC=HW * C)
        y, _, _ = self._conv(o, name + '.proj_out', B, HW, 1, residual=x)
        return y

    # ---- encoder / decoder
    def _encode_f(self, img: torch.Tensor) -> torch.Tensor:
        """quant_conv(encoder(img)) -> f (B, Cvae, 16, 16) fp32 (vqvae.py:74; vae_modules.py:144-160)."""
        P = self._pack()
        if self.encoder_precision != ('bf16' if self.compute_dtype == torch.bfloat16 else 'fp32'):
            return self._encode_f_hiprec(img)
        B, _, H, W = img.shape
        T = self.compute_dtype
        cfg = self.cfg
        cin_p = P['conv']['encoder.conv_in']['cin']
        x = torch.empty(B * H * W, cin_p, device=img.device, dtype=T)
        ops.nchw_to_nhwc(img.contiguous().float(), x, B, 3, H * W, cin_p)
        h, H, W = self._conv(x, 'encoder.conv_in', B, H, W)
        nlev = len(cfg.ch_mult)
        in_mult = (1,) + tuple(cfg.ch_mult)
        cur = cfg.ch
        for lv in range(nlev):
            cout = cfg.ch * cfg.ch_mult[lv]
            for b in range(cfg.num_res_blocks):
                h = self._resblock(h, f'encoder.down.{lv}.block.{b}', B, H, W, cur, cout)
                cur = cout
                if lv == nlev - 1:
                    h = self._attnblock(h, f'encoder.down.{lv}.attn.{b}', B, H * W, cur)
            if lv != nlev - 1:
                h, H, W = self._conv(h, f'encoder.down.{lv}.downsample.conv', B, H, W, stride=2)
        h = self._resblock(h, 'encoder.mid.block_1', B, H, W, cur, cur)
        h = self._attnblock(h, 'encoder.mid.attn_1', B, H * W, cur)
        h = self._resblock(h, 'encoder.mid.block_2', B, H, W, cur, cur)
        h = self._gn(h, 'encoder.norm_out', B, H * W, cur)
        z, _, _ = self._conv(h, 'encoder.conv_out', B, H, W, out_dtype=torch.float32)      # (B*HW, Cvae) fp32 NHWC
        f, _, _ = self._conv(z, 'quant_conv', B, H, W)                                      # fp32 weights
        out = torch.empty(B, self.Cvae, H, W, device=img.device, dtype=torch.float32)
        ops.nhwc_to_nchw(f, self.Cvae, out, B, self.Cvae, H * W)
        return out

    # ---- the encoder in its higher precisions (bf16 model, encoder_precision 'bf16x3' / 'fp32'): fp32 NHWC activations throughout; x3: every conv input is split into
    # [hi | lo | hi] bf16 rows (by the GroupNorm that feeds it, or by cvar_split3) and contracted with [w_hi | w_hi | w_lo] rows on the bf16 tiles.  The 16 x 16 attention
    # blocks (0.1 % of the encoder's flops) run q k^T and p v on the exact-f32 MFMA.  Same layer walk as _encode_f (vae_modules.py:144-160).
    def _hp_conv(self, x, name, B, Hin, Win, *, stride=1, residual=None, pre_split=False):
        c = self._pack()['enc'][name]
        x3mode = self.encoder_precision == 'bf16x3'
        ks = c['ks']
        Hout, Wout = (Hin // 2, Win // 2) if stride == 2 else (Hin, Win)
        M = B * Hout * Wout
        if x3mode and not pre_split:
            xs = torch.empty(B * Hin * Win, c['cin3'], device=x.device, dtype=torch.bfloat16)
            ops.split3(x, xs, B * Hin * Win, c['cin'], c['cin3'])
            x = xs
        cin = c['cin3'] if x3mode else c['cin']
        out = torch.empty(M, c['cout'], device=x.device, dtype=torch.float32)
        if ks == 1:
            ops.gemm(x, c['w'], out, M=M, N=c['cout'], K=cin, bias=c['b'], residual=residual, split_k=False)
        else:
            ops.gemm(x, c['w'], out, M=M, N=c['cout'], K=9 * cin, bias=c['b'], residual=residual,
                     conv=dict(Hin=Hin, Win=Win, Cin=cin, Hout=Hout, Wout=Wout, stride=stride, up=0))
        return out, Hout, Wout

    def _hp_gn(self, x, name, B, HW, C, silu=True):
        """GroupNorm(+SiLU) of the fp32 stream -> the next conv's input: split rows (x3) or fp32"""
        w, b = self._pack()['norm'][name]
        ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=x.device, dtype=torch.uint8)
        if self.encoder_precision == 'bf16x3':
            out = torch.empty(B * HW, 3 * C, device=x.device, dtype=torch.bfloat16)
            return ops.groupnorm_silu_split3(x, w, b, out, B, HW, C, self.cfg.gn_groups, self.cfg.gn_eps, silu, ws)
        return ops.groupnorm_silu(x, w, b, torch.empty_like(x), B, HW, C, self.cfg.gn_groups, self.cfg.gn_eps, silu, ws)

    def _hp_resblock(self, x, name, B, H, W, cin, cout):
        h = self._hp_gn(x, name + '.norm1', B, H * W, cin)
        h, _, _ = self._hp_conv(h, name + '.conv1', B, H, W, pre_split=True)
        h = self._hp_gn(h, name + '.norm2', B, H * W, cout)
        if cin != cout:
            x, _, _ = self._hp_conv(x, name + '.nin_shortcut', B, H, W)
        h, _, _ = self._hp_conv(h, name + '.conv2', B, H, W, residual=x, pre_split=True)
        return h

    def _hp_attnblock(self, x, name, B, HW, C):
        F32 = torch.float32
        n = self._hp_gn(x, name + '.norm', B, HW, C, silu=False)
        qkv, _, _ = self._hp_conv(n, name + '.qkv', B, HW, 1, pre_split=True)             # (B*HW, 3C) fp32: q | k | v
        HWp = -(-HW // 4) * 4
        vT = torch.empty(B, C, HW, device=x.device, dtype=F32) if HWp == HW else torch.zeros(B, C, HWp, device=x.device, dtype=F32)
        ops.transpose(qkv, vT, B, HW, C, 3 * C, in_off=2 * C, ld_out=HWp)
        s = torch.empty(B, HW, HW, device=x.device, dtype=F32)
        ops.gemm(qkv, qkv, s, M=HW, N=HW, K=C, lda=3 * C, ldw=3 * C, w_off=C, alpha=float(int(C) ** -0.5), batch=B,
                 strideA=HW * 3 * C, strideW=HW * 3 * C, strideC=HW * HW)
        p = torch.empty(B, HW, HW, device=x.device, dtype=F32)
        ops.softmax_rows(s, p, B * HW, HW)
        if HWp != HW:
            pp = torch.zeros(B, HW, HWp, device=x.device, dtype=F32)
            pp[:, :, :HW] = p
            p = pp
        o = torch.empty(B * HW, C, device=x.device, dtype=F32)
        ops.gemm(p, vT, o, M=HW, N=C, K=HWp, batch=B, strideA=HW * HWp, strideW=C * HWp, strideC=HW * C)
        y, _, _ = self._hp_conv(o, name + '.proj_out', B, HW, 1, residual=x)
        return y

    def _encode_f_hiprec(self, img: torch.Tensor) -> torch.Tensor:
        P = self._pack()
        B, _, H, W = img.shape
        cfg = self.cfg
        cin_p = P['enc']['encoder.conv_in']['cin']
        x = torch.empty(B * H * W, cin_p, device=img.device, dtype=torch.float32)
        ops.nchw_to_nhwc(img.contiguous().float(), x, B, 3, H * W, cin_p)
        h, H, W = self._hp_conv(x, 'encoder.conv_in', B, H, W)
        nlev = len(cfg.ch_mult)
        cur = cfg.ch
        for lv in range(nlev):
            cout = cfg.ch * cfg.ch_mult[lv]
            for b in range(cfg.num_res_blocks):
                h = self._hp_resblock(h, f'encoder.down.{lv}.block.{b}', B, H, W, cur, cout)
                cur = cout
                if lv == nlev - 1:
                    h = self._hp_attnblock(h, f'encoder.down.{lv}.attn.{b}', B, H * W, cur)
            if lv != nlev - 1:
                h, H, W = self._hp_conv(h, f'encoder.down.{lv}.downsample.conv', B, H, W, stride=2)
        h = self._hp_resblock(h, 'encoder.mid.block_1', B, H, W, cur, cur)
        h = self._hp_attnblock(h, 'encoder.mid.attn_1', B, H * W, cur)
        h = self._hp_resblock(h, 'encoder.mid.block_2', B, H, W, cur, cur)
        h = self._hp_gn(h, 'encoder.norm_out', B, H * W, cur)
        z, _, _ = self._hp_conv(h, 'encoder.conv_out', B, H, W, pre_split=True)             # (B*HW, Cvae) fp32 NHWC
        f, _, _ = self._conv(z, 'quant_conv', B, H, W)                                      # fp32 weights
        out = torch.empty(B, self.Cvae, H, W, device=img.device, dtype=torch.float32)
        ops.nhwc_to_nchw(f, self.Cvae, out, B, self.Cvae, H * W)
        return out

    def _decode(self, f_hat: torch.Tensor, lo=-1.0, hi=1.0, mul=1.0, add=0.0) -> torch.Tensor:
        """decoder(post_quant_conv(f_hat)).clamp(lo,hi)*mul+add (vqvae.py:88-89; vae_modules.py:210-225)."""
        P = self._pack()
        outs = []
        for s in range(0, f_hat.shape[0], self.decode_chunk):
            outs.append(self._decode_chunk(f_hat[s:s + self.decode_chunk].contiguous(), lo, hi, mul, add))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def _decode_chunk(self, f_hat, lo, hi, mul, add):
        cfg = self.cfg
        T = self.compute_dtype
        B, Cv, H, W = f_hat.shape
        x = torch.empty(B * H * W, Cv, device=f_hat.device, dtype=torch.float32)
        ops.nchw_to_nhwc(f_hat.float(), x, B, Cv, H * W, Cv)
        z, _, _ = self._conv(x, 'post_quant_conv', B, H, W, out_dtype=T)                   # fp32 conv, output in compute dtype
        nlev = len(cfg.ch_mult)
        cur = cfg.ch * cfg.ch_mult[-1]
        h, _, _ = self._conv(z, 'decoder.conv_in', B, H, W)
        h = self._resblock(h, 'decoder.mid.block_1', B, H, W, cur, cur)
        h = self._attnblock(h, 'decoder.mid.attn_1', B, H * W, cur)
        h = self._resblock(h, 'decoder.mid.block_2', B, H, W, cur, cur)
        for lv in reversed(range(nlev)):
            cout = cfg.ch * cfg.ch_mult[lv]
            for b in range(cfg.num_res_blocks + 1):
                h = self._resblock(h, f'decoder.up.{lv}.block.{b}', B, H, W, cur, cout)
                cur = cout
                if lv == nlev - 1:
                    h = self._attnblock(h, f'decoder.up.{lv}.attn.{b}', B, H * W, cur)
            if lv != 0:
                h, H, W = self._conv(h, f'decoder.up.{lv}.upsample.conv', B, H, W, up=1)
        h = self._gn(h, 'decoder.norm_out', B, H * W, cur)
        y, _, _ = self._conv(h, 'decoder.conv_out', B, H, W, out_dtype=torch.float32)      # (B*HW, 3) fp32
        out = torch.empty(B, 3, H, W, device=f_hat.device, dtype=torch.float32)
        ops.nhwc_to_nchw(y, 3, out, B, 3, H * W, lo, hi, mul, add)
        return out

    # ---- quantizer helpers
    def _split(self, flat: torch.Tensor, mf: int = 1) -> List[torch.Tensor]:
        outs, o = [], 0
        for pn in self.cfg.patch_nums:
            n = mf * pn * pn
            outs.append(flat[:, o:o + n])
            o += n
        return outs

    def _scale_tables(self, v_patch_nums):
        """operator tables + phi map of a caller-chosen scale list (vqvae.py:73-75 passes v_patch_nums through to
        f_to_idxBl_or_fhat, quant.py:184-215, which accepts any list ending at the latent size; phi by the nearest-tick rule :282-290)"""
        P = self._pack()
        pns = tuple(int(p[0] if isinstance(p, (tuple, list)) else p) for p in v_patch_nums)
        if any(isinstance(p, (tuple, list)) and p[0] != p[1] for p in v_patch_nums):
            raise NotImplementedError('non-square scales (ph != pw) are not built')
        if pns == tuple(self.cfg.patch_nums):
            return pns, P['up'], P['down'], P['phi_map']
        if pns[-1] != self.cfg.patch_nums[-1]:
            raise AssertionError(f'patch_hws[-1]={pns[-1]} != latent size {self.cfg.patch_nums[-1]}')          # quant.py:193
        if len(pns) > 16:
            raise NotImplementedError('at most 16 scales')
        cache = P.setdefault('alt_tables', {})
        if pns not in cache:
            from .spec import phi_index_map
            up, down, _ = packed_tables(pns)
            cache[pns] = (torch.from_numpy(up).to(self.device), torch.from_numpy(down).to(self.device), phi_index_map(len(pns), self.cfg.share_quant_resi))
        up, down, pm = cache[pns]
        return pns, up, down, pm

    def _ms_encode(self, f: torch.Tensor, want_fhat=False, want_margin=False, v_patch_nums=None):
        P = self._pack()
        B = f.shape[0]
        pns, up, down, phi_map = self._scale_tables(v_patch_nums) if v_patch_nums is not None else (self.cfg.patch_nums, P['up'], P['down'], P['phi_map'])
        Ltot = sum(p * p for p in pns)
        idx = torch.empty(B, Ltot, device=f.device, dtype=torch.int32)
        fh = torch.empty_like(f) if want_fhat else None
        mg = torch.empty(B, Ltot, device=f.device, dtype=torch.float32) if want_margin else None
        ops.ms_encode(f.contiguous(), P['E'], self.V, P['phi_w'], P['phi_b'], phi_map, list(pns), up, down, idx, fh, mg,
                      B, pns[-1], self.Cvae)
        return idx, fh, mg

    def _next_input(self, si: int, idx: torch.Tensor, f_hat: torch.Tensor, nb: int, nmaps: int, want_tok: bool, soft: Optional[torch.Tensor] = None,
                    pn_next: Optional[int] = None):
        """one scale step for all (nb, nmaps) maps: f_hat updated in place, returns tokens of the next scale or None.
        soft: (nb, nmaps*pn*pn, Cvae) embeddings used INSTEAD of E[idx] (more_smooth, control_var.py:511-515) - they are handed to the
        kernel as a one-off codebook addressed by the identity index.  pn_next: pool the updated map to this size instead of the next
        scale's (the control pass of separate_decoding feeds area(f_hat -> the SAME scale), control_var.py:467-468)."""
        P = self._pack()
        pns = self.cfg.patch_nums
        pn = pns[si]
        last = si == len(pns) - 1
        explicit = pn_next is not None
        if not explicit:
            pn_next = pns[si + 1] if not last else pn
        codebook = P['E']
        if soft is not None:
            codebook = soft.reshape(-1, self.Cvae).float().contiguous()
            idx = torch.arange(codebook.shape[0], device=f_hat.device, dtype=torch.int32).view(nb, -1)
        make_tok = want_tok and (explicit or not last)
        tok = torch.empty(nb, nmaps * pn_next * pn_next, self.Cvae, device=f_hat.device, dtype=torch.float32) if make_tok else None
        down_off = P['tab_off'][pns.index(pn_next)] if make_tok else 0
        ops.ms_next_input(idx, codebook, P['phi_w'], P['phi_b'], P['up'], P['down'], f_hat, tok, nb, nmaps, pn, pn_next, pns[-1], self.Cvae,
                          P['phi_map'][si], P['tab_off'][si], down_off)
        return tok

    # ---- public API (same names / meaning as models/vqvae.py)
    @torch.no_grad()
    def img_to_idxBl(self, inp_img_no_grad: torch.Tensor, v_patch_nums=None) -> List[torch.Tensor]:
        """vqvae.py:73-75 -> list of (B, pn*pn) int64 ids, coarse to fine"""
        self._pack(check=True)
        idx, _, _ = self._ms_encode(self._encode_f(inp_img_no_grad), v_patch_nums=v_patch_nums)
        pns = self._scale_tables(v_patch_nums)[0] if v_patch_nums is not None else self.cfg.patch_nums
        outs, o = [], 0
        for pn in pns:
            outs.append(idx[:, o:o + pn * pn].long())
            o += pn * pn
        return outs

    @torch.no_grad()
    def idxBl_to_h(self, gt_ms_idx_Bl: List[torch.Tensor]) -> List[torch.Tensor]:
        """vqvae.py:77-78 / quant.py:217-240 -> teacher-forcing inputs, list of (B, pn_{k+1}^2, Cvae) fp32"""
        self._pack(check=True)
        B = gt_ms_idx_Bl[0].shape[0]
        dev = gt_ms_idx_Bl[0].device
        S = self.cfg.patch_nums[-1]
        f_hat = torch.zeros(B, 1, self.Cvae, S, S, device=dev, dtype=torch.float32)
        outs = []
        for si in range(len(self.cfg.patch_nums) - 1):
            outs.append(self._next_input(si, gt_ms_idx_Bl[si].to(torch.int32).contiguous(), f_hat, B, 1, True))
        return outs

    def _idx_to_fhat(self, ms_idx_Bl: List[torch.Tensor]) -> torch.Tensor:
        B = ms_idx_Bl[0].shape[0]
        S = self.cfg.patch_nums[-1]
        f_hat = torch.zeros(B, 1, self.Cvae, S, S, device=ms_idx_Bl[0].device, dtype=torch.float32)
        for si in range(len(self.cfg.patch_nums)):
            self._next_input(si, ms_idx_Bl[si].to(torch.int32).contiguous(), f_hat, B, 1, False)
        return f_hat[:, 0]

    @torch.no_grad()
    def fhat_to_img(self, f_hat: torch.Tensor) -> torch.Tensor:
        """vqvae.py:88-89"""
        self._pack(check=True)
        return self._decode(f_hat)

    @torch.no_grad()
    def idxBl_to_img(self, ms_idx_Bl: List[torch.Tensor], same_shape: bool = True, last_one: bool = False):
        """vqvae.py:97-104; same_shape=False decodes every scale at its own resolution (images of 16 pn x 16 pn)"""
        self._pack(check=True)
        if not same_shape:                                   # every scale at its own resolution (quant.py:171-180)
            rows = [self._embed(idx) for idx in ms_idx_Bl]
            fh = self._lowres_fhats(rows, ms_idx_Bl[0].shape[0], last_one)
            return self._decode(fh) if last_one else [self._decode(f) for f in fh]
        if last_one:
            return self._decode(self._idx_to_fhat(ms_idx_Bl))
        B = ms_idx_Bl[0].shape[0]
        S = self.cfg.patch_nums[-1]
        f_hat = torch.zeros(B, 1, self.Cvae, S, S, device=ms_idx_Bl[0].device, dtype=torch.float32)
        outs = []
        for si in range(len(self.cfg.patch_nums)):
            self._next_input(si, ms_idx_Bl[si].to(torch.int32).contiguous(), f_hat, B, 1, False)
            outs.append(self._decode(f_hat[:, 0]))
        return outs

    def _embed(self, idx_Bl: torch.Tensor) -> torch.Tensor:
        """quantize.embedding(idx) -> (B * l, Cvae) fp32 rows (vqvae.py:103)"""
        P = self._pack()
        _check_index_range(idx_Bl, 0, self.V - 1, 'token ids')           # inclusive bounds: id == V raises like nn.Embedding's IndexError
        out = torch.empty(idx_Bl.numel(), self.Cvae, device=idx_Bl.device, dtype=torch.float32)
        return ops.embed_rows(idx_Bl.to(torch.int32).contiguous(), P['E'], out)

    def _lowres_fhats(self, ms_rows: List[torch.Tensor], B: int, last_one: bool):
        """embed_to_fhat(all_to_max_scale=False) (quant.py:171-180): f_hat starts at the first scale's size, is bicubic-resized to every next
        scale's size and receives phi_k(h_k) computed at THAT resolution.  ms_rows: per scale (B * pn * pn, Cvae) fp32 NHWC rows.
        Returns (B, Cvae, pn, pn) fp32 maps - the last one, or one per scale."""
        from .pyramid import bicubic_matrix
        P = self._pack()
        pns, C, dev = self.cfg.patch_nums, self.Cvae, ms_rows[0].device
        if len(ms_rows) != len(pns):
            raise ValueError(f'expected {len(pns)} scales, got {len(ms_rows)}')
        mats = P.setdefault('lowres_mats', {})
        f, prev, outs = None, None, []
        for si, pn in enumerate(pns):
            n = B * pn * pn
            h = ms_rows[si].reshape(n, C).float().contiguous()
            base = torch.zeros(n, C, device=dev, dtype=torch.float32)
            if f is not None:
                if (prev, pn) not in mats:
                    mats[(prev, pn)] = torch.from_numpy(bicubic_matrix(prev, pn).astype('float32')).to(dev)
                m = mats[(prev, pn)]
                ops.resample_sep(f, m, m, base, B, prev, prev, pn, pn, C)
            k = P['phi_map'][si]
            f = torch.empty(n, C, device=dev, dtype=torch.float32)
            ops.gemm(h, P['phi_fold_w'], f, M=n, N=C, K=9 * C, w_off=k * C * 9 * C, bias=P['phi_fold_b'][k], residual=base,
                     conv=dict(Hin=pn, Win=pn, Cin=C, Hout=pn, Wout=pn))
            prev = pn
            if not last_one or si == len(pns) - 1:
                o = torch.empty(B, C, pn, pn, device=dev, dtype=torch.float32)
                ops.nhwc_to_nchw(f, C, o, B, C, pn * pn)
                outs.append(o)
        return outs[-1] if last_one else outs

    @torch.no_grad()
    def embed_to_fhat(self, ms_h_BChw: List[torch.Tensor], all_to_max_scale: bool = True, last_one: bool = False):
        """quant.py:156-182: multi-scale embeddings (B, Cvae, pn, pn) -> f_hat (the last, or after every scale)"""
        self._pack(check=True)
        B = ms_h_BChw[0].shape[0]
        rows = [h.float().permute(0, 2, 3, 1).reshape(B * h.shape[2] * h.shape[3], self.Cvae).contiguous() for h in ms_h_BChw]
        if not all_to_max_scale:
            return self._lowres_fhats(rows, B, last_one)
        S = self.cfg.patch_nums[-1]
        f_hat = torch.zeros(B, 1, self.Cvae, S, S, device=rows[0].device, dtype=torch.float32)
        outs = []
        for si in range(len(self.cfg.patch_nums)):
            self._next_input(si, None, f_hat, B, 1, False, soft=rows[si].view(B, -1, self.Cvae))
            if not last_one:
                outs.append(f_hat[:, 0].clone())
        return f_hat[:, 0] if last_one else outs

    @torch.no_grad()
    def embed_to_img(self, ms_h_BChw: List[torch.Tensor], all_to_max_scale: bool, last_one: bool = False):
        """vqvae.py:91-95"""
        fh = self.embed_to_fhat(ms_h_BChw, all_to_max_scale=all_to_max_scale, last_one=last_one)
        return self._decode(fh) if last_one else [self._decode(f) for f in fh]

    @torch.no_grad()
    def img_to_recon(self, x, v_patch_nums=None, last_one=False):
        """vqvae.py:80-86"""
        self._pack(check=True)
        idx, fh, _ = self._ms_encode(self._encode_f(x), want_fhat=True, v_patch_nums=v_patch_nums)
        if last_one:
            return self._decode(fh, lo=-3.0e38, hi=3.0e38)
        return self._recon_all(idx, v_patch_nums)

    def _recon_all(self, idx, v_patch_nums=None):
        """decode f_hat after every scale (quant.py:184-215 with to_fhat=True) - for the constructor's scale list or a caller-chosen one"""
        P = self._pack()
        pns, up, down, phi_map = self._scale_tables(v_patch_nums) if v_patch_nums is not None else (self.cfg.patch_nums, P['up'], P['down'], P['phi_map'])
        B = idx.shape[0]
        S = pns[-1]
        f_hat = torch.zeros(B, 1, self.Cvae, S, S, device=idx.device, dtype=torch.float32)
        outs, o, tab = [], 0, 0
        for si, pn in enumerate(pns):
            ids = idx[:, o:o + pn * pn].contiguous()
            ops.ms_next_input(ids, P['E'], P['phi_w'], P['phi_b'], up, down, f_hat, None, B, 1, pn, pn, S, self.Cvae, phi_map[si], tab, 0)
            outs.append(self._decode(f_hat[:, 0], lo=-3.0e38, hi=3.0e38))
            o += pn * pn
            tab += S * pn
        return outs

    def forward(self, *a, **k):
        raise NotImplementedError('VQVAE training (vqvae.py:56-59, losses/) is out of scope: the tokenizer is frozen on the hot path')


# =====================================================================================
# ControlVAR / VAR
# =====================================================================================
FUSE_LN_BELOW = 4097       # passes with fewer rows ask proj / fc2 for the next op's adaLN (see _blocks_and_head); 2048 -> 4097: B = 8 63.0 -> 62.3 ms, B = 16 106.7 -> 106.0


class ControlVAR(nn.Module):
    """Joint (control, image) next-scale transformer (reference: models/control_var.py:23-689).

    Built: aln=1 (AdaLNSABlock), multi_cond as given, and of the non-default variants ``shared_aln``
    and ``type_pos`` for inference, forward and training (both fold into tables at pack time, no extra kernel).
    ``aln < 0`` (SABlock: affine LayerNorms + layer scale, basic_var.py:128-176) is likewise folded into the adaLN layout.
    ``bidirectional`` (image-first order, mask_first=False) swaps the first two tokens and the type ids.
    ``separate_decoding`` (control half of a scale decoded before its image half; with ``indep`` the halves are blind to each other):
    masks as (level end, hole) tables of the attention kernels, the two-pass inference branch, and the mask applied at inference for
    ``indep``.  ``indep`` defaults to True as upstream's class does (a no-op without separate_decoding); the factory passes False.
    ``more_smooth`` = Gumbel-softmax soft code embeddings.  ``separator`` (+18 special tokens, head V + 18, special_embed): upstream
    indexes special_embed with V + k and raises IndexError everywhere (control_var.py:549,606); built with the evidently intended
    index k for forward() / training and the joint inference branch - including upstream's placement quirks there (:507-509,538).
    """
    _control = True

    def __init__(self, vae_local: VQVAE, num_classes=1000, norm_eps=1e-6, aln=1, aln_gamma_init=1e-3, shared_aln=False,
                 cond_drop_rate=0.1, depth=16, embed_dim=1024, num_heads=16, mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., layer_scale=-1., tau=4, cos_attn=False, patch_nums=DEFAULT_PATCH_NUMS,
                 flash_if_available=True, fused_if_available=True, mask_factor=2, bidirectional=False, separate_decoding=False,
                 separator=False, type_pos=False, indep=True, multi_cond=False,
                 compute_dtype=None, init_seed: int = 0, deterministic_plan: bool = False):
        """deterministic_plan (an addition of this library): every transformer GEMM runs on the unsliced tile kernels whatever its row count - no small-M weight-streaming
        kernel, no K slices - so the fp32 summation order of a row does not depend on how many rows ride beside it, and one (label, condition, g_seed) in batch row 0 gives
        the same logits and tokens BIT FOR BIT at any batch size (default False: the small-M plans are 2.0x / 1.24x / 1.05x faster at B = 1 / 8 / 32 and move bf16 logits by ~5e-3 of
        max|logit| between batch sizes; tests/test_gpu_configs.py).  Can also be flipped on a built model: ``model.deterministic_plan = True``."""
        super().__init__()
        self.deterministic_plan = bool(deterministic_plan)
        if separator and not (self._control and mask_factor == 2):
            raise NotImplementedError('separator needs the joint (control, image) sequence')
        if separator and len(patch_nums) != 10:
            raise NotImplementedError('separator: upstream hard-codes 18 special tokens (control_var.py:543)')
        if separate_decoding and not (self._control and mask_factor == 2):
            raise NotImplementedError('separate_decoding needs the joint (control, image) sequence')
        if bidirectional and not (self._control and mask_factor == 2):
            raise NotImplementedError('bidirectional needs the joint (control, image) sequence')
        sa_block = aln < 0                                           # control_var.py:41: using_aln = aln >= 0
        if type_pos and not (self._control and mask_factor == 2):
            raise NotImplementedError('type_pos needs the joint (control, image) sequence: upstream builds type_1L with 2*sum(pn^2) entries')
        if self._control and mask_factor == 2 and not multi_cond:
            raise NotImplementedError('mask_factor=2 requires multi_cond=True (every shipped config)')
        if embed_dim // num_heads != 64:
            raise NotImplementedError('head_dim must be 64')
        self.cfg = VarConfig(depth=depth, mask_factor=mask_factor, multi_cond=bool(multi_cond) and self._control and mask_factor == 2,
                             control=self._control, patch_nums=tuple(patch_nums), vocab=vae_local.vocab_size, cvae=vae_local.Cvae,
                             num_classes=num_classes, embed_dim=embed_dim, num_heads=num_heads, norm_eps=norm_eps, tau=float(tau),
                             cos_attn=bool(cos_attn), mlp_ratio=mlp_ratio, cond_drop_rate=cond_drop_rate, drop_path_rate=float(drop_path_rate),
                             shared_aln=bool(shared_aln) and not sa_block, type_pos=bool(type_pos), sa_block=sa_block,
                             layer_scale=float(layer_scale) if sa_block else -1.0, bidirectional=bool(bidirectional),
                             separate_decoding=bool(separate_decoding), indep=bool(indep) and self._control and mask_factor == 2,
                             separator=bool(separator))
        self.separate_decoding, self.indep, self.separator, self.type_pos = self.cfg.separate_decoding, bool(indep), self.cfg.separator, self.cfg.type_pos
        cfg = self.cfg
        self.bidirectional = cfg.bidirectional
        self.Cvae, self.V = cfg.cvae, cfg.vocab
        self.depth, self.C, self.D, self.num_heads = depth, cfg.C, cfg.C, cfg.H
        if cfg.C > 2048:
            # surfaced here instead of as CVAR_EUNSUPPORTED from inside a GEMM call: the adaLN kernels keep a row of the residual stream in one wave's
            # registers (cvar_ln_modulate / the row-finishing reduction: C <= 2048 = depth <= 32 at 64 channels per head)
            raise ValueError(f'embed_dim {cfg.C} > 2048: the adaLN kernels of controlvar_amd hold one row per wave (depth <= 32)')
        self.patch_nums, self.mask_factor, self.multi_cond = tuple(patch_nums), mask_factor, cfg.multi_cond
        py = cfg.pyramid
        self.L, self.first_l = py.L, py.first_l
        self.begin_ends = list(zip(py.begin, py.end))
        self.num_stages_minus_1 = len(patch_nums) - 1
        self.num_classes = num_classes
        self.cond_drop_rate = cond_drop_rate
        self.prog_si = -1
        self.vae_proxy: Tuple[VQVAE] = (vae_local,)                  # tuple proxy: not a submodule (control_var.py:72-73)
        self.vae_quant_proxy = (vae_local.quantize,)
        self.compute_dtype = _compute_dtype(compute_dtype) if compute_dtype is not None else vae_local.compute_dtype
        _register_tree(self, var_state_shapes(cfg), synth_var_state(cfg, init_seed), requires_grad=True)
        self._packed = None
        self._arena = None
        self.last_trace: Optional[dict] = None

    # ---- plumbing
    def load_state_dict(self, state_dict, strict=True, assign=False):
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._arena = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.pos_1LC.device

    def _state_sig(self):
        return _tensor_sig(self)

    def _matrix_copies(self) -> Tuple[Dict[str, torch.Tensor], Tuple[str, ...]]:
        """(state_dict key -> the contiguous slice of the packed copies that holds exactly this weight matrix, the packed keys those
        slices cover completely).  The fused optimizer writes the rounded update straight into the slices (cvar_adam_tensor.w16)
        and then calls _pack(fresh=keys); bf16 compute only (the fp32 form keeps no separate rounding)."""
        P, cfg = self._pack(), self.cfg
        C, depth = cfg.C, cfg.depth
        if self.compute_dtype != torch.bfloat16:
            return {}, ()
        out: Dict[str, torch.Tensor] = {}
        keys = ['w_qkv', 'w_proj', 'w_fc1', 'w_fc2', 'w_head']
        for i in range(depth):
            for key, name in (('w_qkv', 'attn.mat_qkv.weight'), ('w_proj', 'attn.proj.weight'), ('w_fc1', 'ffn.fc1.weight'), ('w_fc2', 'ffn.fc2.weight')):
                out[f'blocks.{i}.{name}'] = P[key][i]
        if not cfg.sa_block and not cfg.shared_aln:       # the other two forms derive w_ada (zeros / one shared matrix per block)
            for i in range(depth):
                out[f'blocks.{i}.ada_lin.1.weight'] = P['w_ada'][i * 6 * C:(i + 1) * 6 * C]
            out['head_nm.ada_lin.1.weight'] = P['w_ada'][depth * 6 * C:]
            keys.append('w_ada')
        hw = 'head.1.weight' if cfg.sa_block else 'head.weight'
        out[hw] = P['w_head'][:cfg.head_out]
        return out, tuple(keys)

    def _pack(self, check: bool = False, fresh: Sequence[str] = ()):
        """GEMM-ready device copies of the weights (stacked per kind, compute dtype).  check=True (every public entry point and
        the training engine's forward): rebuild when any parameter changed since - torch.optim steps through the autograd bridge,
        manual in-place edits, .data swaps (the copies used to go stale on that path).
        fresh: keys of the current copies that are already up to date (the fused optimizer updated them in its own pass) - those tensors
        are kept, everything else is rebuilt from the parameters."""
        if not fresh and self._packed is not None and (not check or self._packed_sig == self._state_sig()):
            return self._packed
        old = self._packed if fresh else None
        if fresh and old is None:
            raise RuntimeError('_pack(fresh=...) without packed copies to keep')
        dev, T, cfg = self.device, self.compute_dtype, self.cfg
        if dev.type != 'cuda':
            raise RuntimeError('controlvar_amd models compute on the GPU only; call .to("cuda") first')
        ops.ensure_splitk_workspace(dev)
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        C, depth = cfg.C, cfg.depth
        P: Dict[str, Any] = {}
        blk = lambda i, s: sd[f'blocks.{i}.{s}']
        P['w_qkv'] = old['w_qkv'] if 'w_qkv' in fresh else torch.stack([blk(i, 'attn.mat_qkv.weight') for i in range(depth)]).to(T).contiguous()
        P['b_qkv'] = torch.stack([torch.cat((blk(i, 'attn.q_bias'), torch.zeros_like(blk(i, 'attn.q_bias')), blk(i, 'attn.v_bias')))
                                  for i in range(depth)]).float().contiguous()
        P['w_proj'] = old['w_proj'] if 'w_proj' in fresh else torch.stack([blk(i, 'attn.proj.weight') for i in range(depth)]).to(T).contiguous()
        P['b_proj'] = torch.stack([blk(i, 'attn.proj.bias') for i in range(depth)]).float().contiguous()
        P['w_fc1'] = old['w_fc1'] if 'w_fc1' in fresh else torch.stack([blk(i, 'ffn.fc1.weight') for i in range(depth)]).to(T).contiguous()
        P['b_fc1'] = torch.stack([blk(i, 'ffn.fc1.bias') for i in range(depth)]).float().contiguous()
        P['w_fc2'] = old['w_fc2'] if 'w_fc2' in fresh else torch.stack([blk(i, 'ffn.fc2.weight') for i in range(depth)]).to(T).contiguous()
        P['b_fc2'] = torch.stack([blk(i, 'ffn.fc2.bias') for i in range(depth)]).float().contiguous()
        # every ada_lin of the model in ONE weight: rows [i*6C,(i+1)*6C) = block i, last 2C rows = head_nm
        head_w, head_b = ('head.1.weight', 'head.1.bias') if cfg.sa_block else ('head.weight', 'head.bias')
        if cfg.sa_block:
            # SABlock (basic_var.py:128-160) in the adaLN layout: LayerNorm(x) * w + b == LN0(x) * (1 + (w - 1)) + b and the
            # residual gates are the layer-scale gammas (or 1).  The generator weight is zero, its bias holds the constants,
            # so every kernel (and the backward: d bias = column sum of d ada) is shared with the adaLN form.
            one = torch.ones(C, device=dev)
            gam = (lambda i, k: blk(i, k)) if cfg.layer_scale >= 0 else (lambda i, k: one)
            b_blk = [torch.cat([gam(i, 'gamma1'), gam(i, 'gamma2'), blk(i, 'norm1.weight') - 1, blk(i, 'norm2.weight') - 1,
                                blk(i, 'norm1.bias'), blk(i, 'norm2.bias')]) for i in range(depth)]
            w_all = torch.zeros(depth * 6 * C + 2 * C, C, device=dev)
            b_all = torch.cat(b_blk + [sd['head.0.weight'] - 1, sd['head.0.bias']])
        elif cfg.shared_aln:
            # (ada_gss + SharedAdaLin(cond)) of block i (basic_var.py:204-205) == a Linear whose weight is the shared one and whose
            # bias is shared bias + ada_gss_i: replicate the weight per block so the one-GEMM layout below serves both forms
            w_blk = [sd['shared_ada_lin.1.weight']] * depth
            b_blk = [sd['shared_ada_lin.1.bias'] + blk(i, 'ada_gss').reshape(-1) for i in range(depth)]
        else:
            w_blk = [blk(i, 'ada_lin.1.weight') for i in range(depth)]
            b_blk = [blk(i, 'ada_lin.1.bias') for i in range(depth)]
        if not cfg.sa_block:
            w_all = None if 'w_ada' in fresh else torch.cat(w_blk + [sd['head_nm.ada_lin.1.weight']])
            b_all = torch.cat(b_blk + [sd['head_nm.ada_lin.1.bias']])
        P['w_ada'] = old['w_ada'] if 'w_ada' in fresh else w_all.to(T).contiguous()
        P['b_ada'] = b_all.float().contiguous()
        P['n_ada'] = depth * 6 * C + 2 * C
        hw, hb = sd[head_w], sd[head_b].float()
        if cfg.head_ld != hw.shape[0]:               # separator: V + 18 = 4114 columns -> padded to 4120 (zero weight, -1e30 bias: softmax weight exactly 0)
            if 'w_head' not in fresh:
                hw = torch.cat((hw, hw.new_zeros(cfg.head_ld - hw.shape[0], hw.shape[1])))
            hb = torch.cat((hb, hb.new_full((cfg.head_ld - hb.shape[0],), -1e30)))
        P['w_head'] = old['w_head'] if 'w_head' in fresh else hw.to(T).contiguous()
        P['b_head'] = hb.contiguous()
        P['special'] = sd['special_embed.weight'].float().contiguous() if cfg.separator else None
        P['sp_rows'] = {}
        P['w_we'] = sd['word_embed.weight'].float().contiguous()
        P['b_we'] = sd['word_embed.bias'].float().contiguous()
        P['lvl_pos'] = (sd['lvl_embed.weight'][sd['lvl_1L'][0]] + sd['pos_1LC'][0]).float().contiguous()      # (L, C)
        # type_pos (control_var.py:99-117): upstream adds type_embed[type_1L] to every row in forward() (:622-624), to the
        # rows of scales >= 1 in autoregressive_infer_cfg (:423-424,482-483) and nowhere in conditional_infer_cfg
        # the '_' tables are the image-first order (mask_first=False, bidirectional): type ids from type_1L_ (:424,624)
        P['lvl_pos_fwd'] = P['lvl_pos_gen'] = P['lvl_pos_fwd_'] = P['lvl_pos_gen_'] = P['lvl_pos']
        if cfg.type_pos:
            for suf, buf in (('', 'type_1L'), ('_', 'type_1L_')):
                ty = sd['type_embed.weight'][sd[buf][0]].float()
                P['lvl_pos_fwd' + suf] = (P['lvl_pos'] + ty).contiguous()
                P['lvl_pos_gen' + suf] = P['lvl_pos_fwd' + suf].clone()
                P['lvl_pos_gen' + suf][:cfg.pyramid.first_l] = P['lvl_pos'][:cfg.pyramid.first_l]
        P['pos_start'] = sd['pos_start'][0].float().contiguous()
        P['pos_start_sw'] = P['pos_start'].flip(0).contiguous()
        P['class_emb'] = sd['class_emb.weight'].float().contiguous()
        P['cond_embed'] = sd['cond_embed.weight'].float().contiguous() if 'cond_embed.weight' in sd else None
        if cfg.uses_cos_attn:
            P['scale_mul'] = torch.stack([blk(i, 'attn.scale_mul_1H11').reshape(-1) for i in range(depth)]).float().contiguous()
        self._packed = P
        self._packed_sig = self._state_sig()
        self._pack_gen = getattr(self, '_pack_gen', 0) + 1          # consumers of derived copies (TrainEngine.WT) compare this
        return P

    def _first_tokens(self, P, labels, types, x, cond, R: int, x_rows: int, table, mask_first: bool = True):
        """first-scale tokens [cond_token, sos] (+ pos_start + table rows); image first (mask_first=False, control_var.py:404-407,
        587) is [sos, cond_token]: the kernel runs with the two position rows exchanged, then the two token rows are exchanged."""
        py, C = self.cfg.pyramid, self.cfg.C
        if mask_first or py.first_l != 2:
            ops.first_tokens(P['class_emb'], P['cond_embed'], labels, types, P['pos_start'], table, x, cond, R, py.first_l, C, x_rows)
            return
        ops.first_tokens(P['class_emb'], P['cond_embed'], labels, types, P['pos_start_sw'], table[:2].flip(0).contiguous(), x, cond, R,
                         py.first_l, C, x_rows)
        xv = x[:R * x_rows].view(R, x_rows, C)
        xv[:, :2] = xv[:, :2].flip(1)

    def _special_rows(self, table, positions, rows):
        """(len(positions), C): special_embed[rows] + table[positions] - the separator tokens as they enter the residual stream"""
        P = self._pack()
        key = (table.data_ptr(), tuple(positions), tuple(rows))
        if key not in P['sp_rows']:
            P['sp_rows'][key] = (P['special'][list(rows)] + table[list(positions)]).contiguous()
        return P['sp_rows'][key]

    def _embed_teacher_forced(self, P, tok, x, B: int, table, mask_first: bool):
        """word_embed of the teacher-forcing tokens into the rows of x behind the first scale (+ level / position / type rows).  With
        separators the code tokens of every half go to their own row range and the special rows are copied in between
        (control_var.py:603-620, special_embed indexed by the label offset without V)."""
        cfg = self.cfg
        py, C = cfg.pyramid, cfg.C
        if not cfg.separator:
            ops.word_embed(tok, P['w_we'], P['b_we'], table, x, B, 1, py.L - py.first_l, cfg.cvae, C, py.L, py.first_l, lvl_off=py.first_l)
            return
        mapping = cfg.special_mapping(mask_first)
        xv = x[:B * py.L].view(B, py.L, C)
        cur = 0
        for k in range(1, len(py.patch_nums)):
            n = py.patch_nums[k] ** 2
            for h in range(2):
                row0 = py.begin[k] + h * (n + 1)
                ops.word_embed(tok[:, cur:cur + n].contiguous(), P['w_we'], P['b_we'], table, x, B, 1, n, cfg.cvae, C, py.L, row0, lvl_off=row0)
                cur += n
        pos = [int(p) for p in py.special_positions()]
        xv[:, pos] = self._special_rows(table, pos, [mapping[i] for i in range(len(pos))])

    def _get_arena(self, R: int, Lmax: int):
        """KV arena [depth][R][Lmax][2C] (k | v halves; the queries of a scale are dead after its attention and live in a per-call scratch,
        like the reference's cache of k and v only, basic_var.py:108-111) of the calling stream (generations running concurrently on
        different streams must not share it)"""
        sid = torch.cuda.current_stream(self.device).cuda_stream
        key = (R, Lmax, self.compute_dtype)
        if self._arena is None:
            self._arena = {}
        ent = self._arena.get(sid)
        if ent is None or ent[0] != key:
            ent = self._arena[sid] = (key, torch.empty(self.cfg.depth, R, Lmax, 2 * self.cfg.C, device=self.device, dtype=self.compute_dtype))
        return ent[1]

    # ---- one pass of all blocks + head over l new tokens per sequence
    def _blocks_and_head(self, x, ada, R: int, l: int, q_off: int, Lmax: int, arena, lvl_end=None, holes=None):
        """x: (R*l, C) fp32 residual stream (updated in place).  Returns logits (R*l, V) fp32."""
        P, cfg = self._pack(), self.cfg
        C, H, T = cfg.C, cfg.H, self.compute_dtype
        M = R * l
        hid = P['w_fc1'].shape[1]
        n_ada = P['n_ada']
        dev = x.device
        u = torch.empty(M, C, device=dev, dtype=T)
        o = torch.empty(M, C, device=dev, dtype=T)
        hbuf = torch.empty(M, hid, device=dev, dtype=T)
        qs = torch.empty(M, C, device=dev, dtype=T)                 # queries of this pass: (R, l, C)
        arena_stride = R * Lmax * 2 * C
        # bf16 mode: the queries carry softmax scale * log2(e) from the producing epilogue (one rounding of q * c; the attention kernel then
        # spends no multiply per score - cvar_attention_prescaled).  fp32 parity mode: the exact row-wise kernel, unscaled queries.
        LOG2E = 1.4426950408889634
        pre = T == torch.bfloat16
        q_alpha = (float(cfg.attn_scale) * LOG2E) if (pre and not cfg.uses_cos_attn) else 1.0
        # Small passes (early scales, small batches): proj / fc2 also produce the adaLN input of the op that follows (cvar_gemm_desc.ln_out) - their split-K
        # reduction finishes rows, so LayerNorm + modulation ride in that launch instead of a cvar_ln_modulate of their own (same bits).  Large passes keep the
        # separate launch: their GEMMs finish tiles, not rows.
        sm = not self.deterministic_plan      # small-M plans (weight-streaming kernel, K slices): the summation order then depends on M; off = tile kernels only, unsliced
        fuse_ln = M < FUSE_LN_BELOW   # calls up to here can be sliced along K (small-M split-K, the long-K rule of the 256x256 tiles); an unsliced call launches cvar_ln_modulate itself
        ah = cfg.depth * 6 * C
        ops.ln_modulate(x, ada, 2 * C, 4 * C, n_ada, l, u, M, C, cfg.norm_eps)
        for i in range(cfg.depth):
            a0 = i * 6 * C
            # one GEMM for q | k | v: the q columns land in the scratch, k | v rows straight in their KV-arena slots (row remap)
            ops.gemm(u, P['w_qkv'], arena, M=M, N=3 * C, K=C, w_off=i * 3 * C * C, bias=P['b_qkv'][i], c_off=i * arena_stride,
                     ldc=2 * C, remap=(l, Lmax, q_off), split=(qs, C, C), split_alpha=q_alpha, small_m=sm, split_k=sm)
            if cfg.uses_cos_attn:
                ops.cos_qk_norm(arena, R, H, Lmax, q_off, l, P['scale_mul'], qkv_off=i * arena_stride, sm_off=i * H, q=qs, q_mul=LOG2E if pre else 1.0)
            ops.attention(arena, o, R, H, Lmax, q_off, l, float(cfg.attn_scale), lvl_end, qkv_off=i * arena_stride, holes=holes, q=qs, prescaled=pre)
            ln2 = (u, ada, a0 + 3 * C, a0 + 5 * C, n_ada, l, cfg.norm_eps)
            ops.gemm(o, P['w_proj'], x, M=M, N=C, K=C, w_off=i * C * C, bias=P['b_proj'][i], gate=ada, gate_off=a0, ldg=n_ada, gate_rows=l,
                     residual=x, ln=ln2 if fuse_ln else None, small_m=sm, split_k=sm)
            if not fuse_ln:
                ops.ln_modulate(x, *ln2[1:6], u, M, C, cfg.norm_eps)
            ops.gemm(u, P['w_fc1'], hbuf, M=M, N=hid, K=C, w_off=i * hid * C, bias=P['b_fc1'][i], act=ACT_GELU_TANH, small_m=sm, split_k=sm)
            # the row finished by fc2 is the input of the next block's first adaLN (or of the head's)
            a1 = a0 + 6 * C
            ln1 = (u, ada, a1 + 2 * C, a1 + 4 * C, n_ada, l, cfg.norm_eps) if i + 1 < cfg.depth else (u, ada, ah, ah + C, n_ada, l, cfg.norm_eps)
            ops.gemm(hbuf, P['w_fc2'], x, M=M, N=C, K=hid, w_off=i * C * hid, bias=P['b_fc2'][i], gate=ada, gate_off=a0 + C, ldg=n_ada,
                     gate_rows=l, residual=x, ln=ln1 if fuse_ln else None, small_m=sm, split_k=sm)
            if not fuse_ln:
                ops.ln_modulate(x, *ln1[1:6], u, M, C, cfg.norm_eps)
        logits = torch.empty(M, cfg.head_ld, device=dev, dtype=torch.float32)
        ops.gemm(u, P['w_head'], logits, M=M, N=cfg.head_ld, K=C, bias=P['b_head'], small_m=sm, split_k=sm)
        return logits

    def _ada(self, cond: torch.Tensor, R: int):
        P = self._pack()
        cs = torch.empty(R, self.cfg.C, device=cond.device, dtype=self.compute_dtype)
        ops.silu_cast(cond, cs)
        ada = torch.empty(R, P['n_ada'], device=cond.device, dtype=torch.float32)
        sm = not self.deterministic_plan
        ops.gemm(cs, P['w_ada'], ada, M=R, N=P['n_ada'], K=self.cfg.C, bias=P['b_ada'], small_m=sm, split_k=sm)
        return ada

    def _as_labels(self, B, label_B, seed):
        dev = self.device
        if label_B is None:
            g = torch.Generator(device='cpu').manual_seed(seed)
            label_B = torch.randint(0, self.num_classes, (B,), generator=g)
        elif isinstance(label_B, int):
            label_B = torch.full((B,), self.num_classes if label_B < 0 else label_B)
        _check_index_range(label_B, 0, self.num_classes, 'label_B')
        return label_B.to(device=dev, dtype=torch.int32)

    def _as_types(self, B, cond_type, seed, allow_none=True):
        dev = self.device
        if cond_type is None:
            if B == 4:
                cond_type = torch.tensor([0, 1, 2, 3])                      # control_var.py:387-389
            else:
                g = torch.Generator(device='cpu').manual_seed(seed + 1)
                cond_type = torch.randint(0, 4, (B,), generator=g)
        elif isinstance(cond_type, int):
            assert 0 < cond_type <= 3                                      # control_var.py:395
            cond_type = torch.full((B,), cond_type)
        _check_index_range(cond_type, 0, 4, 'cond_type')
        return cond_type.to(device=dev, dtype=torch.int32)

    @torch.no_grad()
    def _prepare_rows(self, B, label_B, cond_type, four_way, seed):
        """labels / condition types of all CFG rows: [cond ; uncond] or the 4-branch layout (control_var.py:252-263,381-400)"""
        cfg = self.cfg
        labels = self._as_labels(B, label_B, seed)
        empty = torch.full_like(labels, self.num_classes)
        nrep = 4 if four_way else 2
        labels_all = torch.cat([labels] + [empty] * (nrep - 1)).contiguous()
        types_all = None
        if cfg.mask_factor == 2:
            types = self._as_types(B, cond_type, seed)
            e4 = torch.full_like(types, 4)
            types_all = (torch.cat([types, types, e4, e4]) if four_way else torch.cat([types, e4])).contiguous()
        return labels_all, types_all

    @torch.no_grad()
    def _generate(self, B, label_B, g_seed, cfg_scale, top_k, top_p, more_smooth, cond_type, four_way, c_mask, c_img,
                  force_idx=None, trace: bool = False, gumbel=None):
        if top_k > self.cfg.vocab:                     # helpers.py:8-10: torch.topk raises on k > V; top_k <= 0 means no top-k filter
            raise RuntimeError(f'selected index k out of range (top_k={top_k} > vocabulary {self.cfg.vocab})')
        seed = int(g_seed) if g_seed is not None else int(torch.empty((), dtype=torch.int64).random_().item())
        self._pack(check=True); self.vae_proxy[0]._pack(check=True)
        for name, ids in (('c_mask', c_mask), ('c_img', c_img), ('_force_idx', force_idx)):
            if ids is not None:
                _check_index_range(torch.cat([t.reshape(-1) for t in ids]), 0, self.cfg.vocab - 1, name)
        labels_all, types_all = self._prepare_rows(B, label_B, cond_type, four_way, seed)
        mask_first = True
        if self.cfg.mask_factor == 2 and not four_way:
            # control_var.py:403: python's global `random`, drawn on every call (before the `or`), exactly as upstream -
            # random.seed(k) before the call reproduces the reference's choice of order
            import random
            mask_first = True if (random.random() < 0.5 or not self.bidirectional) else False
        if self.cfg.separator and (four_way or more_smooth or (self.cfg.separate_decoding and not self.cfg.indep)):
            raise NotImplementedError('separator: conditional_infer_cfg ignores the special tokens (control_var.py:270-330), the two-pass branch fails with a '
                                      'shape error (:481) and more_smooth slices the soft embeddings at shifted positions upstream; only forward(), training '
                                      'and the joint autoregressive_infer_cfg branch are built')
        if self.cfg.separate_decoding and not self.cfg.indep and not four_way:              # control_var.py:428-485
            return self._generate_two_pass(B, labels_all, types_all, seed, cfg_scale, top_k, top_p, bool(more_smooth), force_idx, trace, mask_first, gumbel)
        return self._generate_core(B, labels_all, types_all, seed, None, cfg_scale, top_k, top_p, four_way, c_mask, c_img, force_idx, trace,
                                   mask_first=mask_first, more_smooth=bool(more_smooth), gumbel=gumbel)

    @torch.no_grad()
    def _generate_core(self, B, labels_all, types_all, seed, seed_dev, cfg_scale, top_k, top_p, four_way, c_mask=None, c_img=None,
                       force_idx=None, trace: bool = False, mask_first: bool = True, more_smooth: bool = False, gumbel=None):
        """the 10-scale loop on device-resident inputs only (capturable in a HIP graph: no host sync, static shapes)"""
        cfg, P = self.cfg, self._pack()
        vae: VQVAE = self.vae_proxy[0]
        py, mf, C = cfg.pyramid, cfg.mask_factor, cfg.C
        dev = self.device
        nrep = 4 if four_way else 2
        R = nrep * B
        nb = R if four_way else B
        x = torch.empty(R * py.l[-1], C, device=dev, dtype=torch.float32)
        cond = torch.empty(R, C, device=dev, dtype=torch.float32)
        self._first_tokens(P, labels_all, types_all, x, cond, R, py.first_l, P['lvl_pos'], mask_first)
        gen_table = P['lvl_pos'] if four_way else (P['lvl_pos_gen'] if mask_first else P['lvl_pos_gen_'])
        ada = self._ada(cond, R)
        arena = self._get_arena(R, py.L)
        S = py.patch_nums[-1]
        f_hat = torch.zeros(nb, mf, cfg.cvae, S, S, device=dev, dtype=torch.float32)
        tr = {'idx': [], 'margin': [], 'logits': []} if trace else None
        nstage = len(py.patch_nums)
        # `indep` models hand the rows of the training mask to every inference pass (control_var.py:283,497); those rows hide something
        # only under separate_decoding (otherwise every cached key is visible anyway)
        inf_lvl, inf_holes = attention_levels(cfg) if (cfg.indep and cfg.separate_decoding) else (None, None)
        Pv = vae._pack()
        for si, pn in enumerate(py.patch_nums):
            l = py.l[si]
            ratio = si / (nstage - 1)
            logits = self._blocks_and_head(x[:R * l], ada, R, l, py.begin[si], py.L, arena, lvl_end=inf_lvl, holes=inf_holes)
            if four_way:
                t1, t2, t3 = [c * ratio for c in cfg_scale]
                coef = [1 + t1, t2 - t1, t3 - t2, -t3]
            else:
                t = cfg_scale * ratio
                coef = [1 + t, -t]
            n_draw = 4 if four_way else 1
            idx = torch.empty(n_draw * B, l, device=dev, dtype=torch.int32)
            comb = torch.empty(B, l, cfg.vocab, device=dev, dtype=torch.float32) if trace else None
            mg = torch.empty(B, l, device=dev, dtype=torch.float32) if trace else None
            soft = None
            if more_smooth and top_k != 1:          # greedy: the in-place masked softmax is one-hot, i.e. exactly E[idx] (control_var.py:511-515)
                soft = torch.empty(n_draw * B, l, cfg.cvae, device=dev, dtype=torch.float32)
                ops.cfg_sample(logits, B, nrep, l, cfg.vocab, coef, top_k, top_p, seed, si, n_draw, idx, comb, mg, seed_dev=seed_dev, codebook=Pv['E'],
                               smooth_mul=1.0 + ratio, smooth_tau=max(0.27 * (1 - ratio * 0.95), 0.005),
                               gumbel=gumbel[si].to(device=dev, dtype=torch.float32).contiguous() if gumbel is not None else None, soft_out=soft,
                               ldv=cfg.head_ld)
            else:
                ops.cfg_sample(logits, B, nrep, l, cfg.vocab, coef, top_k, top_p, seed, si, n_draw, idx, comb, mg, seed_dev=seed_dev, ldv=cfg.head_ld)
            if trace:
                tr['idx'].append(idx.clone()); tr['margin'].append(mg); tr['logits'].append(comb)
            if force_idx is not None:
                idx = force_idx[si].to(device=dev, dtype=torch.int32).contiguous()
            if four_way:                                                 # teacher forcing (control_var.py:309-324)
                if c_mask is not None:
                    idx[:3 * B, :pn * pn] = c_mask[si].to(device=dev, dtype=torch.int32).repeat(3, 1)
                if c_img is not None:
                    idx[:3 * B, pn * pn:] = c_img[si].to(device=dev, dtype=torch.int32).repeat(3, 1)
            if cfg.separator and py.sp(si):
                # control_var.py:507-509: the ids drawn at the separator positions are dropped from the THIRD scale on; at the second scale
                # upstream keeps all 10 ids and slices [:pn^2] / [-pn^2:], so the image half becomes ids 6..9 - replicated literally
                n = pn * pn
                idx = (torch.cat((idx[:, :n], idx[:, n + 1:2 * n + 1]), dim=1) if si > 1 else torch.cat((idx[:, :n], idx[:, -n:]), dim=1)).contiguous()
            tok = vae._next_input(si, idx, f_hat, nb, mf, True, soft=soft)
            if si != nstage - 1 and cfg.separator:
                # control_var.py:536-552: upstream stacks the two next-scale maps along H and splits the stack at row `pn` (the CURRENT
                # scale's size) before putting a separator behind each part: the first one lands behind pn * pn' tokens
                ln, pn2 = py.l[si + 1], py.patch_nums[si + 1]
                cut, nrep_x = pn * pn2, 2
                ops.word_embed(tok[:, :cut].contiguous(), P['w_we'], P['b_we'], gen_table, x, nb, nrep_x, cut, cfg.cvae, C, ln, 0, lvl_off=py.end[si])
                ops.word_embed(tok[:, cut:].contiguous(), P['w_we'], P['b_we'], gen_table, x, nb, nrep_x, 2 * pn2 * pn2 - cut, cfg.cvae, C, ln, cut + 1,
                               lvl_off=py.end[si] + cut + 1)
                mapping = cfg.special_mapping(mask_first)
                x[:R * ln].view(R, ln, C)[:, [cut, ln - 1]] = self._special_rows(gen_table, [py.end[si] + cut, py.end[si] + ln - 1], [mapping[2 * si], mapping[2 * si + 1]])
            elif si != nstage - 1:
                ln = py.l[si + 1]
                ops.word_embed(tok, P['w_we'], P['b_we'], gen_table, x, nb, 1 if four_way else 2, ln,
                               cfg.cvae, C, ln, 0, lvl_off=py.end[si])
        if trace:
            tr['f_hat'] = f_hat[:B].clone()
            self.last_trace = tr
        return f_hat[:B]

    @torch.no_grad()
    def _generate_two_pass(self, B, labels_all, types_all, seed, cfg_scale, top_k, top_p, more_smooth, force_idx, trace, mask_first, gumbel):
        """separate_decoding without indep (control_var.py:428-485): 2 x 10 passes over the same KV arena - per scale first the control
        tokens, then the image tokens, which see their scale's control tokens through the cache (attn_bias=None upstream).  The inputs
        cross over as upstream's do: the image pass of scale k is fed the CONTROL f_hat pooled to pn_k (:467-468), the control pass of
        scale k+1 the IMAGE f_hat pooled to pn_{k+1} (:470)."""
        cfg, P = self.cfg, self._pack()
        if cfg.type_pos:
            raise NotImplementedError('type_pos + separate_decoding: upstream indexes patch_nums[si + 1] with si up to 18 (control_var.py:483) and raises')
        vae: VQVAE = self.vae_proxy[0]
        Pv = vae._pack()
        py, C, dev = cfg.pyramid, cfg.C, self.device
        pns = py.patch_nums
        R = 2 * B
        xf = torch.empty(R * 2, C, device=dev, dtype=torch.float32)
        cond = torch.empty(R, C, device=dev, dtype=torch.float32)
        self._first_tokens(P, labels_all, types_all, xf, cond, R, 2, P['lvl_pos'], mask_first)
        ada = self._ada(cond, R)
        arena = self._get_arena(R, py.L)
        S = pns[-1]
        f = [torch.zeros(B, 1, cfg.cvae, S, S, device=dev, dtype=torch.float32) for _ in range(2)]
        x = torch.empty(R * S * S, C, device=dev, dtype=torch.float32)
        tr = {'idx': [], 'margin': [], 'logits': []} if trace else None
        nstage, pos = len(pns), 0
        for si in range(2 * nstage):
            half, k = si % 2, si // 2
            pn = pns[k]
            l = pn * pn
            ratio = k / (nstage - 1)
            if si < 2:
                x[:R].copy_(xf.view(R, 2, C)[:, si])
            logits = self._blocks_and_head(x[:R * l], ada, R, l, pos, py.L, arena)
            pos += l
            t = cfg_scale * ratio
            idx = torch.empty(B, l, device=dev, dtype=torch.int32)
            comb = torch.empty(B, l, cfg.vocab, device=dev, dtype=torch.float32) if trace else None
            mg = torch.empty(B, l, device=dev, dtype=torch.float32) if trace else None
            soft = None
            if more_smooth and top_k != 1:
                soft = torch.empty(B, l, cfg.cvae, device=dev, dtype=torch.float32)
                ops.cfg_sample(logits, B, 2, l, cfg.vocab, [1 + t, -t], top_k, top_p, seed, si, 1, idx, comb, mg, codebook=Pv['E'], smooth_mul=1.0 + ratio,
                               smooth_tau=max(0.27 * (1 - ratio * 0.95), 0.005),
                               gumbel=gumbel[si].to(device=dev, dtype=torch.float32).contiguous() if gumbel is not None else None, soft_out=soft)
            else:
                ops.cfg_sample(logits, B, 2, l, cfg.vocab, [1 + t, -t], top_k, top_p, seed, si, 1, idx, comb, mg)
            if trace:
                tr['idx'].append(idx.clone()); tr['margin'].append(mg); tr['logits'].append(comb)
            if force_idx is not None:
                idx = force_idx[si].to(device=dev, dtype=torch.int32).contiguous()
            if si == 2 * nstage - 1:
                vae._next_input(k, idx, f[1], B, 1, False, soft=soft)
                break
            tok = vae._next_input(k, idx, f[half], B, 1, True, soft=soft, pn_next=pn if half == 0 else None)
            ln = tok.shape[1]
            ops.word_embed(tok, P['w_we'], P['b_we'], P['lvl_pos'], x, B, 2, ln, cfg.cvae, C, ln, 0, lvl_off=pos)
        f_hat = torch.cat(f, dim=1)
        if trace:
            tr['f_hat'] = f_hat.clone()
            self.last_trace = tr
        return f_hat

    @torch.no_grad()
    def graphed_generator(self, B: int, cfg=1.5, top_k: int = 0, top_p: float = 0.0):
        """Capture one full `autoregressive_infer_cfg` (10 scales x depth blocks + both VQVAE decodes, ~2.5k launches) in a
        HIP graph and return ``run(label_B, cond_type=None, g_seed=None) -> images``.  Labels, condition types and the
        sampling seed live in static device buffers that are refreshed before each replay, so every call draws new samples.
        Removes the host launch cost that dominates small batches (the reference's loop is host-launched op by op)."""
        dev = self.device
        four_way = False
        if self.bidirectional:
            raise NotImplementedError('the captured generator fixes the (control, image) order; bidirectional models draw it per call')
        lab0 = torch.zeros(B, dtype=torch.int64)
        ty0 = torch.zeros(B, dtype=torch.int64) if self.cfg.mask_factor == 2 else None
        labels_all, types_all = self._prepare_rows(B, lab0, ty0, four_way, 0)
        seed_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        self._pack(); self.vae_proxy[0]._pack()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the capture (module loads, attribute setup, arena)
            self._decode_pair(self._generate_core(B, labels_all, types_all, 0, seed_dev, cfg, top_k, top_p, four_way))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self._arena:                                     # the warm-up's K/V arena (arenas are per stream): the capture allocates its own
            self._arena.pop(side.cuda_stream, None)         # in the graph's pool, and two of them do not fit at large batches
        ops.release_splitk_workspace(dev, side.cuda_stream)  # ... and the warm-up stream's split-K workspace
        torch.cuda.empty_cache()
        ws_before, arena_before = ops.splitk_workspace_keys(), set(self._arena or ())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._decode_pair(self._generate_core(B, labels_all, types_all, 0, seed_dev, cfg, top_k, top_p, four_way))
        # buffers created during the capture live in the GRAPH's memory pool and are baked into its launches: they belong to the graph, not
        # to the per-stream tables - a later stream that is handed the capture stream's (pooled, reused) handle must not inherit them
        owned = [ops.take_splitk_workspace(k) for k in ops.splitk_workspace_keys() - ws_before]
        owned += [self._arena.pop(k) for k in set(self._arena or ()) - arena_before]

        def run(label_B, cond_type=None, g_seed=None):
            seed = int(g_seed) if g_seed is not None else int(torch.empty((), dtype=torch.int64).random_().item())
            la, ta = self._prepare_rows(B, label_B, cond_type, four_way, seed)
            labels_all.copy_(la)
            if types_all is not None:
                types_all.copy_(ta)
            seed_dev.fill_(seed & (2 ** 62 - 1))
            graph.replay()
            return out.clone()

        run.graph = graph
        run.owned = owned                                   # dropped together with the graph when `run` goes away
        return run

    def _decode_pair(self, f_hat: torch.Tensor) -> torch.Tensor:
        vae: VQVAE = self.vae_proxy[0]
        B, mf = f_hat.shape[:2]
        if mf == 1:
            return vae._decode(f_hat[:, 0].contiguous(), lo=-1.0, hi=1.0, mul=0.5, add=0.5)
        # control and image maps go through the decoder as ONE batch (b-major: [b][map]): at small B every decoder launch is a handful of workgroups
        # and latency-bound, so two passes of B images cost twice one pass of 2 B (B = 1: 78 conv + 234 GroupNorm launches -> 39 + 117).  Per-image
        # results do not depend on the batch they ride in (tests/test_gpu_parity.py::test_batch_rows_are_independent...), so the pixels are unchanged.
        img = vae._decode(f_hat.reshape(B * mf, *f_hat.shape[2:]), lo=-1.0, hi=1.0, mul=0.5, add=0.5)
        H, W = img.shape[-2:]
        return img.view(B, mf, 3, H, W).permute(0, 2, 1, 3, 4).reshape(B, 3, mf * H, W)          # control on top, RGB below (control_var.py:563-565)

    # ---- public API
    @torch.no_grad()
    def autoregressive_infer_cfg(self, B: int, label_B, g_seed: Optional[int] = None, cfg=1.5, top_k=0, top_p=0.0,
                                 more_smooth=False, cond_type=None, _force_idx=None, _trace=False, _gumbel=None) -> torch.Tensor:
        """control_var.py:356-565: returns (B, 3, 512, 256) in [0,1] (control image on top, RGB below).
        more_smooth: Gumbel-softmax soft code embeddings (:511-515); `_gumbel` (tests) injects the per-pass noise instead of drawing it."""
        f_hat = self._generate(B, label_B, g_seed, cfg, top_k, top_p, more_smooth, cond_type, False, None, None, _force_idx, _trace, _gumbel)
        return self._decode_pair(f_hat)

    @torch.no_grad()
    def conditional_infer_cfg(self, B: int, label_B, g_seed: Optional[int] = None, cfg=(1.5, 1.5, 1.5), top_k=0, top_p=0.0,
                              more_smooth=False, cond_type=None, c_mask=None, c_img=None, _force_idx=None, _trace=False, _gumbel=None) -> torch.Tensor:
        """control_var.py:223-354: 4-branch CFG with teacher forcing of the control (c_mask) or image (c_img) ids."""
        if self.mask_factor != 2:
            raise NotImplementedError('conditional_infer_cfg needs mask_factor == 2 (control_var.py:333)')
        f_hat = self._generate(B, label_B, g_seed, tuple(cfg), top_k, top_p, more_smooth, cond_type, True, c_mask, c_img, _force_idx, _trace, _gumbel)
        return self._decode_pair(f_hat)

    def forward(self, label_B: torch.LongTensor, x_BLCv_wo_first_l: torch.Tensor, cond_type=None, mask_first=True) -> torch.Tensor:
        """control_var.py:568-651 teacher-forced logits (B, L, V) fp32.  Under autograd (grad mode on and trainable
        parameters) the call is differentiable - `loss.backward()` runs the hand-written backward kernels
        (controlvar_amd/train.py); otherwise it is the inference-only fast path.  Label / cond-type dropout (rate ``cond_drop_rate``,
        torch.rand) is applied on EVERY call, in train and eval mode alike, exactly as the reference's forward does
        (control_var.py:578,584); DropPath follows ``self.training`` (helpers.py:39-46)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .train import teacher_forced_with_grad
            return teacher_forced_with_grad(self, label_B, x_BLCv_wo_first_l, cond_type, bool(mask_first))
        with torch.no_grad():
            return self._forward_nograd(label_B, x_BLCv_wo_first_l, cond_type, mask_first)

    def _forward_nograd(self, label_B, x_BLCv_wo_first_l, cond_type=None, mask_first=True):
        cfg, P = self.cfg, self._pack(check=True)
        py, C = cfg.pyramid, cfg.C
        dev = self.device
        mask_first = bool(mask_first) or cfg.mask_factor != 2
        B = x_BLCv_wo_first_l.shape[0]
        _check_index_range(label_B, 0, self.num_classes, 'label_B')
        labels = label_B.to(dev)
        if cfg.cond_drop_rate > 0:                  # control_var.py:578,584: applied on every forward(), train or eval mode
            labels = torch.where(torch.rand(B, device=dev) < cfg.cond_drop_rate, self.num_classes, labels)
        types = None
        if cfg.mask_factor == 2:
            _check_index_range(cond_type, 0, 4, 'cond_type')
            types = cond_type.to(dev)
            if cfg.cond_drop_rate > 0:
                types = torch.where(torch.rand(B, device=dev) < cfg.cond_drop_rate, 4, types)
            types = types.to(torch.int32).contiguous()
        labels = labels.to(torch.int32).contiguous()
        x = torch.empty(B * py.L, C, device=dev, dtype=torch.float32)
        cond = torch.empty(B, C, device=dev, dtype=torch.float32)
        table = P['lvl_pos_fwd'] if mask_first else P['lvl_pos_fwd_']
        self._first_tokens(P, labels, types, x, cond, B, py.L, table, mask_first)
        tok = x_BLCv_wo_first_l.to(device=dev, dtype=torch.float32).contiguous()
        if tok.shape[1] != len(py.code_positions()) - py.first_l:
            raise AssertionError(f'teacher-forcing input has {tok.shape[1]} tokens, expected {len(py.code_positions()) - py.first_l}')       # control_var.py:617
        self._embed_teacher_forced(P, tok, x, B, table, mask_first)
        ada = self._ada(cond, B)
        arena = self._get_arena(B, py.L)
        lvl_end, holes = attention_levels(cfg)
        logits = self._blocks_and_head(x, ada, B, py.L, 0, py.L, arena, lvl_end=lvl_end, holes=holes)
        return logits.view(B, py.L, cfg.head_ld)[:, :, :cfg.head_out]


class VAR(ControlVAR):
    """Plain class-conditional VAR (reference: models/var.py:20-291): mask_factor 1, L = 680."""
    _control = False

    def __init__(self, vae_local: VQVAE, num_classes=1000, norm_eps=1e-6, aln=1, aln_gamma_init=1e-3, shared_aln=False,
                 cond_drop_rate=0.1, depth=16, embed_dim=1024, num_heads=16, mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., layer_scale=-1., tau=4, cos_attn=False, patch_nums=DEFAULT_PATCH_NUMS,
                 flash_if_available=True, fused_if_available=True, compute_dtype=None, init_seed: int = 0, deterministic_plan: bool = False):
        super().__init__(vae_local, num_classes=num_classes, norm_eps=norm_eps, aln=aln, aln_gamma_init=aln_gamma_init,
                         shared_aln=shared_aln, cond_drop_rate=cond_drop_rate, depth=depth, embed_dim=embed_dim, num_heads=num_heads,
                         mlp_ratio=mlp_ratio, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         layer_scale=layer_scale, tau=tau, cos_attn=cos_attn, patch_nums=patch_nums,
                         flash_if_available=flash_if_available, fused_if_available=fused_if_available, mask_factor=1,
                         multi_cond=False, compute_dtype=compute_dtype, init_seed=init_seed, deterministic_plan=deterministic_plan)

    @torch.no_grad()
    def autoregressive_infer_cfg(self, B: int, label_B, g_seed: Optional[int] = None, cfg=1.5, top_k=0, top_p=0.0,
                                 more_smooth=False, _force_idx=None, _trace=False) -> torch.Tensor:
        """var.py:143-207: returns (B, 3, 256, 256) in [0,1]."""
        f_hat = self._generate(B, label_B, g_seed, cfg, top_k, top_p, more_smooth, None, False, None, None, _force_idx, _trace)
        return self._decode_pair(f_hat)

    def conditional_infer_cfg(self, *a, **k):
        raise NotImplementedError('plain VAR has no conditional_infer_cfg (var.py)')

    def forward(self, label_B, x_BLCv_wo_first_l, cond_type=None, mask_first=True):
        return super().forward(label_B, x_BLCv_wo_first_l, None, True)


# =====================================================================================
# factories (models/__init__.py:6-45)
# =====================================================================================
def build_vae(vocab_size=4096, z_channels=32, ch=160, share_quant_resi=4, v_patch_nums=DEFAULT_PATCH_NUMS, test_mode=True, **kw) -> VQVAE:
    """The VQVAE every script builds (train_control_var_hpu.py:581-582)."""
    return VQVAE(vocab_size=vocab_size, z_channels=z_channels, ch=ch, test_mode=test_mode, share_quant_resi=share_quant_resi,
                 v_patch_nums=v_patch_nums, **kw)


def build_var(vae: VQVAE, depth: int, patch_nums=DEFAULT_PATCH_NUMS, aln=1, aln_gamma_init=1e-3, shared_aln=False, layer_scale=-1,
              tau=4, cos_attn=False, flash_if_available=True, fused_if_available=True, **kw) -> VAR:
    return VAR(vae_local=vae, patch_nums=patch_nums, depth=depth, embed_dim=depth * 64, num_heads=depth, drop_path_rate=0.1 * depth / 24,
               aln=aln, aln_gamma_init=aln_gamma_init, shared_aln=shared_aln, layer_scale=layer_scale, tau=tau, cos_attn=cos_attn,
               flash_if_available=flash_if_available, fused_if_available=fused_if_available, **kw)


def build_control_var(vae: VQVAE, depth: int, patch_nums=DEFAULT_PATCH_NUMS, aln=1, aln_gamma_init=1e-3, shared_aln=False, layer_scale=-1,
                      tau=4, cos_attn=False, flash_if_available=True, fused_if_available=True, mask_type='replace', cond_drop_rate=0.1,
                      bidirectional=False, separate_decoding=False, separator=False, type_pos=False, indep=False, multi_cond=False,
                      **kw) -> ControlVAR:
    if mask_type == 'replace':
        mask_factor = 1
    elif mask_type == 'interleave_append':
        mask_factor = 2
    else:
        raise NotImplementedError
    return ControlVAR(vae_local=vae, patch_nums=patch_nums, depth=depth, embed_dim=depth * 64, num_heads=depth,
                      drop_path_rate=0.1 * depth / 24, aln=aln, aln_gamma_init=aln_gamma_init, shared_aln=shared_aln,
                      layer_scale=layer_scale, tau=tau, cos_attn=cos_attn, cond_drop_rate=cond_drop_rate,
                      flash_if_available=flash_if_available, fused_if_available=fused_if_available, mask_factor=mask_factor,
                      bidirectional=bidirectional, separate_decoding=separate_decoding, separator=separator, type_pos=type_pos,
                      indep=indep, multi_cond=multi_cond, **kw)
