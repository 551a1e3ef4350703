#!/usr/bin/env python3
"""Per-scale attention micro-benchmark in the INFERENCE form (d24 geometry: H=24, head_dim 64, R = 2 x batch rows, K/V arena rows of Lmax=1360 +
the scale's queries in their own buffer), interleaved A/B of
    v1    the round-2 MFMA kernel (cvar_attention_v1)
    v2    the round-3 kernel behind cvar_attention (V by transpose-reads, XCD-aware block ids, idle waves skip the math)
    v2p   cvar_attention_prescaled (v2 + the maximum subtracted on the matrix pipe, no multiply per score)
Usage: attn_bench.py [batch=128] [rounds=3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd.spec import VarConfig
dev = torch.device('cuda:0'); T = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = VarConfig(depth=24); py = cfg.pyramid
R, H, L, C = 2 * B, cfg.H, py.L, cfg.C
kv = (torch.randn(R, L, 2 * C, device=dev) * 0.5).to(T)
scale = 0.03125
variants = {'v1': dict(v1=True), 'v2': dict(), 'v2p': dict(prescaled=True)}
tot = {k: 0.0 for k in variants}
tot_fl = 0.0
for s, (b, e) in enumerate(zip(py.begin, py.end)):
    l = e - b
    q = (torch.randn(R * l, C, device=dev) * 0.5).to(T)
    qp = (q.float() * (scale * 1.4426950408889634)).to(T)
    out = torch.empty(R * l, C, device=dev, dtype=T)
    best = {k: 1e9 for k in variants}
    for rd in range(rounds):
        for name, kw in variants.items():
            fn = lambda: ops.attention(kv, out, R, H, L, b, l, scale, None, q=qp if name == 'v2p' else q, **kw)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / 5)
    fl = 4.0 * R * H * l * e * 64
    tot_fl += fl
    for k in variants: tot[k] += best[k]
    print(f'scale {s}: l={l:4d} kv={e:5d}  ' + '  '.join(f'{k} {best[k]:7.4f} ms {fl / best[k] / 1e9:6.1f} TF/s' for k in variants), flush=True)
print(f'all scales (B={B}, R={R}): ' + '  '.join(f'{k} {tot[k]:.3f} ms per layer {tot_fl / tot[k] / 1e9:.1f} TF/s' for k in variants))
