#!/usr/bin/env python3
"""Randomised sweep of the device resampler (cvar_resample_u8 through preprocess.resize_u8) against Pillow's Image.resize:
random source / target sizes (up- and down-scaling, extreme aspect ratios, 1-pixel axes), 1 or 3 channels, LANCZOS and BICUBIC.
The claim under test is bit-identity.  usage: fuzz_resample.py [n_cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from controlvar_amd import preprocess

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    pick = lambda: rng.choice([1, 2, 3, 7, 16, 31, 64, 100, 255, 256, 288, 333, 500, 512, 777, 1024])
    h, w, oh, ow = pick(), pick(), pick(), pick()
    c = rng.choice([1, 3, 3])        # RGBA is out of scope: Pillow premultiplies alpha around the resample, the reference only resizes RGB
    filt = rng.choice(['lanczos', 'bicubic'])
    kind = rng.choice(['noise', 'smooth', 'edges'])
    g = np.random.default_rng(case)
    if kind == 'noise':
        img = g.integers(0, 256, (h, w, c), dtype=np.uint8)
    elif kind == 'smooth':
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(127 + 120 * np.sin(xx / (3 + k) + yy / (5 + k))).astype(np.uint8) for k in range(c)], axis=-1)
    else:
        img = (g.integers(0, 2, (h, w, c)) * 255).astype(np.uint8)
    mode = {1: 'L', 3: 'RGB'}[c]
    pil = Image.fromarray(img[..., 0] if c == 1 else img, mode)
    want = np.asarray(pil.resize((ow, oh), Image.LANCZOS if filt == 'lanczos' else Image.BICUBIC)).reshape(oh, ow, c)
    got = preprocess.resize_u8(torch.from_numpy(img).to(dev), oh, ow, filt).cpu().numpy()
    if not np.array_equal(got, want):
        bad += 1
        d = np.abs(got.astype(int) - want.astype(int))
        print('FAIL', case, dict(h=h, w=w, oh=oh, ow=ow, c=c, filt=filt, kind=kind), 'max diff', d.max(), 'n diff', int((d > 0).sum()), flush=True)
print(f'{n_cases - bad}/{n_cases} cases ok')
sys.exit(1 if bad else 0)
