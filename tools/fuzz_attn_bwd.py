#!/usr/bin/env python3
"""Randomised sweep of the MFMA attention backward (dq and dk/dv kernels) against the exact row-wise backward of the same
library: random rows / heads / block-causal level structures (ragged level lengths), qkv, o, do and dqkv embedded in
NaN-filled buffers so that a read or write outside a tensor shows up.  usage: fuzz_attn_bwd.py [n_cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops

dev = torch.device('cuda:0'); T = torch.bfloat16
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
PAD = 4096


def arena(shape, fill=None, gen=None, amp=1.0):
    n = 1
    for s in shape:
        n *= s
    buf = torch.full((n + 2 * PAD,), float('nan'), device=dev, dtype=T)
    v = buf[PAD:PAD + n].view(*shape)
    if gen is not None:
        v.copy_((torch.randn(*shape, generator=gen) * amp).to(T))
    elif fill is not None:
        v.fill_(fill)
    return buf, v


bad = 0
for case in range(n_cases):
    R, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 4, 12])
    nl = rng.randint(1, 10)
    ends, acc = [], 0
    for _ in range(nl):
        acc += rng.choice([1, 2, 7, 8, 18, 50, 64, 72, 128, 200, 338, 512]); ends.append(acc)
    if acc > 1400:
        ends = [e for e in ends if e <= 1400] or [1400]
        acc = ends[-1]
    L = acc
    C3 = 3 * H * 64
    g = torch.Generator().manual_seed(case)
    amp = rng.choice([0.3, 1.0, 2.0])
    scale = rng.choice([0.125, 0.03125, 0.5])
    _, qkv = arena((R, L, C3), gen=g, amp=amp)
    _, do = arena((R * L, H * 64), gen=g, amp=1.0)
    ob, out = arena((R * L, H * 64), fill=0.0)
    lse = torch.empty(R, H, L, device=dev, dtype=torch.float32)
    ops.attention(qkv, out, R, H, L, 0, L, scale, ends, lse=lse)
    ws = torch.empty(R * H * L + 64, device=dev)
    b1, d1 = arena((R, L, C3), fill=0.0)
    b2, d2 = arena((R, L, C3), fill=0.0)
    ops.attention_bwd(qkv, out, do, lse, d1, ws, R, H, L, L, scale, ends)
    ops.attention_bwd(qkv, out, do, lse, d2, ws, R, H, L, L, scale, ends, rowwise=True)
    a, b = d1.float(), d2.float()
    fin = torch.isfinite(a).all() and torch.isfinite(b).all()
    pads_ok = all(torch.isnan(x[:PAD]).all() and torch.isnan(x[-PAD:]).all() for x in (b1, b2, ob))
    ref = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item() / ref if fin else float('nan')
    ok = fin and pads_ok and err < 3e-2
    if not ok:
        bad += 1
        print('FAIL', case, dict(R=R, H=H, L=L, ends=ends, scale=scale, amp=amp), 'err', err, 'finite', bool(fin), 'pads intact', bool(pads_ok), flush=True)
print(f'{n_cases - bad}/{n_cases} cases ok')
sys.exit(1 if bad else 0)
