"""Host-side operator tables of the token pyramid (area-down / bicubic-up matrices).

These are the published semantics of ``F.interpolate(mode='area'|'bicubic')`` as the
reference calls them (models/quant.py:199,209,235,238,254,256), built in float64 and
rounded once to fp32; the HIP kernels apply them separably from these tables.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def area_matrix(src: int, dst: int) -> np.ndarray:
    """(dst, src): adaptive average pooling bins [floor(i*src/dst), ceil((i+1)*src/dst))."""
    m = np.zeros((dst, src), dtype=np.float64)
    for i in range(dst):
        s = (i * src) // dst
        e = -((-(i + 1) * src) // dst)
        m[i, s:e] = 1.0 / (e - s)
    return m


def bicubic_matrix(src: int, dst: int, A: float = -0.75) -> np.ndarray:
    """(dst, src): cubic convolution, align_corners=False, border taps clamped."""
    m = np.zeros((dst, src), dtype=np.float64)
    scale = src / dst
    for i in range(dst):
        x = (i + 0.5) * scale - 0.5
        x0 = math.floor(x)
        t = x - x0
        w = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
             ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A]
        for k in range(4):
            j = min(max(x0 - 1 + k, 0), src - 1)
            m[i, j] += w[k]
    return m


def packed_tables(patch_nums: Sequence[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """up (S x pn) and down (pn x S) tables packed back to back in pyramid order + offsets
    (the layout cvar_ms_encode / cvar_ms_next_input expect)."""
    S = patch_nums[-1]
    ups, downs, offs = [], [], []
    o = 0
    for pn in patch_nums:
        ups.append(bicubic_matrix(pn, S).astype(np.float32).reshape(-1))
        downs.append(area_matrix(S, pn).astype(np.float32).reshape(-1))
        offs.append(o)
        o += S * pn
    return np.concatenate(ups), np.concatenate(downs), np.asarray(offs, dtype=np.int64)
