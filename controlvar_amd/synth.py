"""Deterministic synthetic weights and inputs (no checkpoints / datasets are reachable here).

Every tensor is generated from (seed, crc32(key)) with a per-key scale chosen so that
activations stay O(1) through the conv / transformer stacks and the logits have a usable
top-1 margin (needed for token-exact greedy parity, SURVEY.md section 7 "Hard parts").
The same recipe feeds the reference (in tests/golden/make_golden.py), the CPU oracle and
the HIP path, so only seeds - never weights - are committed.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch

from .spec import VaeConfig, VarConfig, vae_state_shapes, var_state_shapes


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFFFFFF)
    return g


def _randn(shape, g, std=1.0, mean=0.0):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean


def synth_var_state(cfg: VarConfig, seed: int = 0, head_gain: float = 4.0) -> Dict[str, torch.Tensor]:
    """state_dict for ControlVAR / VAR with the key set of spec.var_state_shapes()."""
    C = cfg.C
    py = cfg.pyramid
    out: Dict[str, torch.Tensor] = {}
    for key, (shape, kind) in var_state_shapes(cfg).items():
        g = _gen(seed, key)
        if key == 'lvl_1L':
            out[key] = torch.from_numpy(py.level_of_token()).view(1, -1)
        elif key == 'attn_bias_for_masking':
            from .spec import attention_bias_matrix
            vis = torch.from_numpy(attention_bias_matrix(cfg))
            out[key] = torch.where(vis, 0.0, -torch.inf).reshape(1, 1, py.L, py.L).contiguous()
        elif key in ('type_1L', 'type_1L_'):
            first = 1 if key == 'type_1L' else 0                     # control half id; the image half gets the other one
            ids = []
            for k, pn in enumerate(py.patch_nums):
                ids += [first] * (pn * pn + py.sp(k)) + [1 - first] * (pn * pn + py.sp(k))
            out[key] = torch.tensor(ids, dtype=torch.int64).view(1, -1)
        elif key.endswith('gamma1') or key.endswith('gamma2'):
            out[key] = _randn(shape, g, std=0.1, mean=0.35)
        elif len(shape) == 1 and key.endswith('.weight'):            # LayerNorm weights of the SABlock variant
            out[key] = _randn(shape, g, std=0.1, mean=1.0)
        elif key.endswith('ada_gss'):
            b = _randn(shape, g, std=0.1)
            b[0, 0, :2] += 0.35
            out[key] = b
        elif key.endswith('zero_k_bias'):
            out[key] = torch.zeros(shape)
        elif key.endswith('scale_mul_1H11'):
            out[key] = _randn(shape, g, std=0.3, mean=math.log(4.0))
        elif key in ('pos_start', 'pos_1LC', 'lvl_embed.weight', 'cond_embed.weight', 'type_embed.weight', 'special_embed.weight'):
            out[key] = _randn(shape, g, std=0.5)
        elif key == 'class_emb.weight':
            out[key] = _randn(shape, g, std=1.0)
        elif key.endswith('ada_lin.1.weight'):
            out[key] = _randn(shape, g, std=0.4 / math.sqrt(C))
        elif key.endswith('ada_lin.1.bias'):
            b = _randn(shape, g, std=0.1)
            if key.startswith('blocks.') and not cfg.shared_aln:
                b[:2 * C] += 0.35            # gamma1, gamma2 rows: residual gates around 0.35
            out[key] = b
        elif key in ('head.weight', 'head.1.weight'):
            out[key] = _randn(shape, g, std=head_gain / math.sqrt(C))
        elif key.endswith('.weight') and len(shape) == 2:
            out[key] = _randn(shape, g, std=1.0 / math.sqrt(shape[1]))
        elif key.endswith('bias') or key.endswith('q_bias') or key.endswith('v_bias'):
            out[key] = _randn(shape, g, std=0.05)
        else:
            raise KeyError(key)
        assert tuple(out[key].shape) == tuple(shape), key
    return out


def synth_vae_state(cfg: VaeConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """state_dict for VQVAE with the key set of spec.vae_state_shapes()."""
    out: Dict[str, torch.Tensor] = {}
    for key, (shape, kind) in vae_state_shapes(cfg).items():
        g = _gen(seed, key)
        if key == 'quantize.ema_vocab_hit_SV':
            out[key] = torch.zeros(shape)
        elif key == 'quantize.embedding.weight':
            out[key] = _randn(shape, g, std=0.6)
        elif '.norm' in key and key.endswith('.weight'):
            out[key] = _randn(shape, g, std=0.1, mean=1.0)
        elif '.norm' in key and key.endswith('.bias'):
            out[key] = _randn(shape, g, std=0.1)
        elif key.endswith('.weight') and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if key.startswith('quantize.quant_resi'):
                gain = 0.7
            elif 'conv_out' in key:
                gain = 1.5
            out[key] = _randn(shape, g, std=gain / math.sqrt(fan_in))
        elif key.endswith('.bias'):
            out[key] = _randn(shape, g, std=0.05)
        else:
            raise KeyError(key)
        assert tuple(out[key].shape) == tuple(shape), key
    return out


def synth_images(batch: int, size: int = 256, seed: int = 0) -> torch.Tensor:
    """(B,3,size,size) fp32 in [-1,1]: smooth low-frequency field + uniform noise
    (SURVEY.md section 8(c) G1 / 8(d) configs 1,5)."""
    g = _gen(seed, f'images{size}')
    low = torch.rand((batch, 3, 8, 8), generator=g) * 2 - 1
    smooth = torch.nn.functional.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
    noise = torch.rand((batch, 3, size, size), generator=g) * 2 - 1
    return (0.75 * smooth + 0.25 * noise).clamp_(-1, 1).contiguous()


# ---- synthetic "decoded JPEG" + segmentation-style condition for the input-pipeline tests (tests/golden/preprocess.npz)
PREPROC_CASES = [(375, 500, 0), (500, 333, 1), (288, 300, 2), (120, 90, 3), (1024, 683, 4)]     # (h, w, seed)


def synth_photo_pair(h: int, w: int, seed: int):
    """uint8 (h, w, 3) image (smooth field + noise) and a 512x512x3 condition (flat-colour discs on black), numpy only"""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(yy / 17.0 + seed) * np.cos(xx / 23.0) + rng.normal(0, 25, (h, w))
    img = np.clip(np.stack([base, base[::-1], base[:, ::-1]], -1), 0, 255).astype(np.uint8)
    cond = np.zeros((512, 512, 3), np.uint8)
    for k in range(4):
        cy, cx, r = rng.integers(60, 450, 3)
        r = int(r) // 4 + 20
        m = (np.mgrid[0:512, 0:512][0] - cy) ** 2 + (np.mgrid[0:512, 0:512][1] - cx) ** 2 < r * r
        cond[m] = rng.integers(0, 5, 3) * 64
    return img, cond


def synth_annotations(seed: int, n: int = 6, size: int = 512):
    """SAM-style segmentation annotations (imagenetC.py:15-29 consumes a list of these): n ellipses on a size x size canvas, each as
    {'area', 'segmentation': {'size', 'counts': uncompressed column-major run lengths}, '_mask': the decoded (size, size) uint8 mask}.
    Shared by tests/golden/make_golden.py (which feeds the decoded masks to the reference's own process_anns) and the tests."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    anns = []
    for _ in range(n):
        cy, cx = rng.integers(40, size - 40, 2)
        ry, rx = rng.integers(10, 140, 2)
        m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1).astype(np.uint8)
        flat = m.T.reshape(-1)                                       # column-major
        change = np.flatnonzero(np.diff(flat)) + 1
        edges = np.concatenate([[0], change, [flat.size]])
        runs = np.diff(edges).tolist()
        if flat[0] == 1:
            runs = [0] + runs
        anns.append({'area': int(m.sum()), 'segmentation': {'size': [size, size], 'counts': runs}, '_runs': runs, '_mask': m})
    return anns
