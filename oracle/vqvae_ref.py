"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the VQVAE multi-scale tokenizer of lxa9867/ControlVAR, written from the
math, operating directly on a state_dict of fp32 tensors.

Pinned against the reference itself: tests/golden/make_golden.py imports
/root/reference/models in the build container, runs both on the same seeded weights/inputs
and records the reference's outputs as fixtures (tests/golden/*.npz); tests/test_oracle_*.py
replay those fixtures against this file.  (The reference has no tests of its own)

``prec`` selects the rounding model: Prec(False) = the reference's fp32 CPU path;
Prec(True) = the bf16 storage points of the HIP bf16 path (fp32 accumulate), used only to
check the GPU bf16 mode.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .interp import area_matrix, bicubic_matrix

SD = Dict[str, torch.Tensor]


class Prec:
    """Rounding model: ``act`` marks every tensor the HIP bf16 path stores in bf16."""

    def __init__(self, bf16: bool = False, q_round: str = 'prescaled'):
        # q_round (bf16 only): where the attention queries are rounded.  'prescaled' = bf16(q * scale * log2 e), the HIP path's storage
        # point since round 3; 'plain' = bf16(q), then the scale in fp32 - the rounding point of the reference's autocast and of rounds 1-2.
        # Both are the same function up to one bf16 rounding of q; tests hold the HIP logits against BOTH so that a wrong prescale constant
        # cannot be absorbed by an oracle that mirrors it.
        assert q_round in ('prescaled', 'plain')
        self.bf16, self.q_round = bf16, q_round

    def act(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.bf16 else x

    w = act


FP32 = Prec(False)


# ----------------------------------------------------------------------------- conv stack
def _conv(sd: SD, name: str, x, prec: Prec, stride=1, padding=1):
    return F.conv2d(x, prec.w(sd[name + '.weight']), sd[name + '.bias'], stride=stride, padding=padding)


def _gn_silu(sd: SD, name: str, x, prec: Prec, silu=True, groups=32, eps=1e-6):
    """GroupNorm(32, eps 1e-6, affine) [+ SiLU]  (vae_modules.py:18-19,58-59)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = xg.var(dim=2, unbiased=False, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    y = y * sd[name + '.weight'].view(1, C, 1, 1) + sd[name + '.bias'].view(1, C, 1, 1)
    if silu:
        y = y * torch.sigmoid(y)
    return prec.act(y)


def _resblock(sd: SD, name: str, x, prec: Prec):
    """vae_modules.py:57-60."""
    h = prec.act(_conv(sd, name + '.conv1', _gn_silu(sd, name + '.norm1', x, prec), prec))
    h = _conv(sd, name + '.conv2', _gn_silu(sd, name + '.norm2', h, prec), prec)
    if (name + '.nin_shortcut.weight') in sd:
        x = prec.act(_conv(sd, name + '.nin_shortcut', x, prec, padding=0))
    return prec.act(x + h)


def _attnblock(sd: SD, name: str, x, prec: Prec):
    """Single-head attention over H*W tokens, scale C^-0.5 (vae_modules.py:73-92)."""
    B, C, H, W = x.shape
    qkv = prec.act(_conv(sd, name + '.qkv', _gn_silu(sd, name + '.norm', x, prec, silu=False), prec, padding=0))
    q, k, v = qkv.reshape(B, 3, C, H * W).unbind(1)              # each (B, C, HW)
    s = torch.bmm(q.transpose(1, 2), k) * (int(C) ** -0.5)         # (B, HWq, HWk)
    if prec.bf16:
        m = s.amax(dim=2, keepdim=True)
        p = torch.exp(s - m)
        o = torch.bmm(prec.act(p), v.transpose(1, 2)) / p.sum(dim=2, keepdim=True)   # (B, HWq, C)
    else:
        o = torch.bmm(F.softmax(s, dim=2), v.transpose(1, 2))
    h = prec.act(o).transpose(1, 2).reshape(B, C, H, W)
    return prec.act(x + _conv(sd, name + '.proj_out', h, prec, padding=0))


def encoder(sd: SD, img: torch.Tensor, prec: Prec = FP32, nlev: int = 5, nres: int = 2) -> torch.Tensor:
    """Encoder.forward (vae_modules.py:144-160): (B,3,256,256) -> (B,Cvae,16,16)."""
    h = prec.act(_conv(sd, 'encoder.conv_in', prec.act(img), prec))
    for lv in range(nlev):
        for b in range(nres):
            h = _resblock(sd, f'encoder.down.{lv}.block.{b}', h, prec)
            if f'encoder.down.{lv}.attn.{b}.norm.weight' in sd:
                h = _attnblock(sd, f'encoder.down.{lv}.attn.{b}', h, prec)
        if lv != nlev - 1:
            h = prec.act(_conv(sd, f'encoder.down.{lv}.downsample.conv', F.pad(h, (0, 1, 0, 1)), prec, stride=2, padding=0))
    h = _resblock(sd, 'encoder.mid.block_1', h, prec)
    h = _attnblock(sd, 'encoder.mid.attn_1', h, prec)
    h = _resblock(sd, 'encoder.mid.block_2', h, prec)
    h = _gn_silu(sd, 'encoder.norm_out', h, prec)
    return _conv(sd, 'encoder.conv_out', h, prec)     # fp32 out


def decoder(sd: SD, z: torch.Tensor, prec: Prec = FP32, nlev: int = 5, nres: int = 2) -> torch.Tensor:
    """Decoder.forward (vae_modules.py:210-225): (B,Cvae,16,16) -> (B,3,256,256)."""
    h = prec.act(_conv(sd, 'decoder.conv_in', prec.act(z), prec))
    h = _resblock(sd, 'decoder.mid.block_1', h, prec)
    h = _attnblock(sd, 'decoder.mid.attn_1', h, prec)
    h = _resblock(sd, 'decoder.mid.block_2', h, prec)
    for lv in reversed(range(nlev)):
        for b in range(nres + 1):
            h = _resblock(sd, f'decoder.up.{lv}.block.{b}', h, prec)
            if f'decoder.up.{lv}.attn.{b}.norm.weight' in sd:
                h = _attnblock(sd, f'decoder.up.{lv}.attn.{b}', h, prec)
        if lv != 0:
            h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)      # nearest x2
            h = prec.act(_conv(sd, f'decoder.up.{lv}.upsample.conv', h, prec))
    h = _gn_silu(sd, 'decoder.norm_out', h, prec)
    return _conv(sd, 'decoder.conv_out', h, prec)     # fp32 out


def img_to_f(sd: SD, img, prec: Prec = FP32) -> torch.Tensor:
    """quant_conv(encoder(img))  (vqvae.py:74).  quant_conv runs fp32 in both modes."""
    return F.conv2d(encoder(sd, img, prec), sd['quant_conv.weight'], sd['quant_conv.bias'], padding=1)


def fhat_to_img(sd: SD, f_hat, prec: Prec = FP32) -> torch.Tensor:
    """decoder(post_quant_conv(f_hat)).clamp(-1,1)  (vqvae.py:88-89)."""
    z = F.conv2d(f_hat, sd['post_quant_conv.weight'], sd['post_quant_conv.bias'], padding=1)
    return decoder(sd, z, prec).clamp(-1, 1)


# ------------------------------------------------------------------ multi-scale quantizer
class MSQuant:
    """VectorQuantizer2 inference helpers (quant.py:156-260), always fp32."""

    def __init__(self, sd: SD, patch_nums: Sequence[int], phi_map: Sequence[int]):
        self.E = sd['quantize.embedding.weight'].float()           # (V, Cvae)
        self.pn = tuple(patch_nums)
        self.S = self.pn[-1]
        self.phi_map = list(phi_map)
        self.phi_w = [sd[f'quantize.quant_resi.qresi_ls.{k}.weight'] for k in sorted(set(phi_map))]
        self.phi_b = [sd[f'quantize.quant_resi.qresi_ls.{k}.bias'] for k in sorted(set(phi_map))]
        self.A = {p: torch.from_numpy(area_matrix(self.S, p)).float() for p in self.pn}       # (p, S)
        self.Bc = {p: torch.from_numpy(bicubic_matrix(p, self.S)).float() for p in self.pn}   # (S, p)

    # -- operators
    def area(self, f: torch.Tensor, p: int) -> torch.Tensor:
        """(B,C,S,S) -> (B,C,p,p) adaptive average (identity at p == S)."""
        if p == self.S:
            return f
        A = self.A[p]
        return torch.einsum('ih,bchw,jw->bcij', A, f, A)

    def bicubic_up(self, h: torch.Tensor) -> torch.Tensor:
        """(B,C,p,p) -> (B,C,S,S) (identity at p == S)."""
        p = h.shape[-1]
        if p == self.S:
            return h
        M = self.Bc[p]
        return torch.einsum('ih,bchw,jw->bcij', M, h, M)

    def phi(self, si: int, h: torch.Tensor, ratio: float = 0.5) -> torch.Tensor:
        """Phi.forward (quant.py:269-270): (1-r)*h + r*conv3x3(h)."""
        k = self.phi_map[si]
        return h * (1 - ratio) + F.conv2d(h, self.phi_w[k], self.phi_b[k], padding=1) * ratio

    def nearest(self, z_NC: torch.Tensor, return_margin: bool = False):
        """argmin_v |z|^2 + |e_v|^2 - 2 z.e_v, first minimum (quant.py:204-206)."""
        d = z_NC.square().sum(dim=1, keepdim=True) + self.E.square().sum(dim=1)
        d = d - 2.0 * (z_NC @ self.E.t())
        idx = torch.argmin(d, dim=1)
        if return_margin:
            top2 = torch.topk(d, 2, dim=1, largest=False).values
            return idx, (top2[:, 1] - top2[:, 0])
        return idx

    def embed(self, idx_Bl: torch.Tensor, p: int) -> torch.Tensor:
        """ids (B, p*p) -> (B, Cvae, p, p)."""
        B = idx_Bl.shape[0]
        return self.E[idx_Bl].transpose(1, 2).reshape(B, -1, p, p)

    # -- A12: f_to_idxBl_or_fhat (quant.py:184-215)
    def f_to_idx(self, f: torch.Tensor, to_fhat: bool = False, return_margins: bool = False):
        B, C = f.shape[:2]
        f_rest = f.clone()
        f_hat = torch.zeros_like(f)
        out, margins = [], []
        for si, p in enumerate(self.pn):
            z = self.area(f_rest, p).permute(0, 2, 3, 1).reshape(-1, C)
            if return_margins:
                idx, mg = self.nearest(z, True)
                margins.append(mg.reshape(B, p * p))
            else:
                idx = self.nearest(z)
            idx = idx.reshape(B, p * p)
            h = self.phi(si, self.bicubic_up(self.embed(idx, p)))
            f_hat = f_hat + h
            f_rest = f_rest - h
            out.append(f_hat.clone() if to_fhat else idx)
        return (out, margins) if return_margins else out

    # -- A13: idxBl_to_var_input (quant.py:217-240)
    def idx_to_var_input(self, ms_idx: List[torch.Tensor]) -> List[torch.Tensor]:
        B = ms_idx[0].shape[0]
        f_hat = torch.zeros(B, self.E.shape[1], self.S, self.S)
        outs = []
        for si in range(len(self.pn) - 1):
            f_hat = f_hat + self.phi(si, self.bicubic_up(self.embed(ms_idx[si], self.pn[si])))
            nxt = self.pn[si + 1]
            outs.append(self.area(f_hat, nxt).reshape(B, -1, nxt * nxt).transpose(1, 2))
        return outs

    # -- A14: get_next_autoregressive_input (quant.py:243-260); functional (returns new f_hat)
    def next_input(self, si: int, f_hat: torch.Tensor, h: torch.Tensor):
        f_hat = f_hat + self.phi(si, self.bicubic_up(h))
        if si != len(self.pn) - 1:
            return f_hat, self.area(f_hat, self.pn[si + 1])
        return f_hat, f_hat

    # -- A18: embed_to_fhat(all_to_max_scale=True, last_one=True) (quant.py:156-170)
    def idx_to_fhat(self, ms_idx: List[torch.Tensor]) -> torch.Tensor:
        B = ms_idx[0].shape[0]
        f_hat = torch.zeros(B, self.E.shape[1], self.S, self.S)
        for si, p in enumerate(self.pn):
            f_hat = f_hat + self.phi(si, self.bicubic_up(self.embed(ms_idx[si], p)))
        return f_hat


def embed_to_fhat_lowres(msq: MSQuant, ms_h: List[torch.Tensor]) -> List[torch.Tensor]:
    """quantize.embed_to_fhat(all_to_max_scale=False, last_one=False) (quant.py:171-180): f_hat grows with the scales - bicubic-resized to
    each scale's size, then phi_k(h_k) added at that resolution.  Returns the list of f_hat after every scale."""
    B, C = ms_h[0].shape[:2]
    f_hat = torch.zeros(B, C, msq.pn[0], msq.pn[0])
    outs = []
    for si, p in enumerate(msq.pn):
        if f_hat.shape[-1] != p:
            M = torch.from_numpy(bicubic_matrix(f_hat.shape[-1], p)).float()
            f_hat = torch.einsum('ih,bchw,jw->bcij', M, f_hat, M)
        f_hat = f_hat + msq.phi(si, ms_h[si])
        outs.append(f_hat)
    return outs


def idxBl_to_img_lowres(sd: SD, msq: MSQuant, ms_idx, prec: Prec = FP32) -> List[torch.Tensor]:
    """VQVAE.idxBl_to_img(same_shape=False, last_one=False) (vqvae.py:97-104): one image per scale, 16 pn pixels wide"""
    ms_h = [msq.embed(idx, p) for idx, p in zip(ms_idx, msq.pn)]
    return [fhat_to_img(sd, f, prec) for f in embed_to_fhat_lowres(msq, ms_h)]


def img_to_idxBl(sd: SD, msq: MSQuant, img, prec: Prec = FP32):
    """VQVAE.img_to_idxBl (vqvae.py:73-75)."""
    return msq.f_to_idx(img_to_f(sd, img, prec))


def idxBl_to_img(sd: SD, msq: MSQuant, ms_idx, prec: Prec = FP32):
    """VQVAE.idxBl_to_img(same_shape=True, last_one=True) (vqvae.py:97-104)."""
    return fhat_to_img(sd, msq.idx_to_fhat(ms_idx), prec)
