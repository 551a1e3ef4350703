"""ORACLE (test infrastructure only - never imported by the product path).

Area-downsample and bicubic-upsample operators of the multi-scale quantizer, restated as
explicit matrices.

The reference calls ``F.interpolate(mode='area')`` (models/quant.py:199,238,256) and
``F.interpolate(mode='bicubic')`` (models/quant.py:209,235,254), i.e. torch's ATen kernels
(torch pinned 2.2.2 in requirements.txt:2; this container has 2.10.0).  Their published
semantics, restated here and checked against torch in tests/test_oracle_interp.py:

* area  == adaptive average pooling with bins [floor(i*S/p), ceil((i+1)*S/p))
* bicubic (align_corners=False, no antialias) == separable cubic convolution with A=-0.75,
  source coordinate x=(i+0.5)*p/S-0.5, taps floor(x)-1..floor(x)+2 clamped to [0,p-1].
"""
from __future__ import annotations

import math

import numpy as np


def area_matrix(src: int, dst: int) -> np.ndarray:
    """(dst, src) float64: row i averages source bins [floor(i*src/dst), ceil((i+1)*src/dst))."""
    m = np.zeros((dst, src), dtype=np.float64)
    for i in range(dst):
        s = (i * src) // dst
        e = -((-(i + 1) * src) // dst)
        m[i, s:e] = 1.0 / (e - s)
    return m


def _cubic_w(t: float, A: float = -0.75):
    def c1(u):  # |u| <= 1
        return ((A + 2.0) * u - (A + 3.0)) * u * u + 1.0

    def c2(u):  # 1 < |u| < 2
        return ((A * u - 5.0 * A) * u + 8.0 * A) * u - 4.0 * A

    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def bicubic_matrix(src: int, dst: int) -> np.ndarray:
    """(dst, src) float64 upsampling operator (identity when src == dst)."""
    m = np.zeros((dst, src), dtype=np.float64)
    scale = src / dst
    for i in range(dst):
        x = (i + 0.5) * scale - 0.5
        x0 = math.floor(x)
        t = x - x0
        for k, w in enumerate(_cubic_w(t)):
            j = min(max(x0 - 1 + k, 0), src - 1)
            m[i, j] += w
    return m
