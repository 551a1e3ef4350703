#!/usr/bin/env python3
"""Attention backward of one d24 training step layer (R = 32 sequences, 24 heads, L = 1360, the block-causal level mask of 'interleave_append'): ms per call of the
whole backward (prep + dQ + dK/dV) by HIP events.  A/B of kernel variants: CVAR_LIB=ab/libcvar_<tag>.so.  usage: attn_bwd_bench.py [R=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN
dev = torch.device('cuda:0'); T = torch.bfloat16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, scale = 24, 0.03125
ends, acc = [], 0
for p in PN:
    acc += 2 * p * p; ends.append(acc)
L = acc
g = torch.Generator().manual_seed(0)
qkv = torch.randn(R, L, 3 * H * 64, generator=g).to(T).to(dev)
do = torch.randn(R * L, H * 64, generator=g).to(T).to(dev)
out = torch.empty(R * L, H * 64, device=dev, dtype=T)
lse = torch.empty(R, H, L, device=dev, dtype=torch.float32)
ops.attention(qkv, out, R, H, L, 0, L, scale, ends, lse=lse)
ws = torch.empty(R * H * L + 64, device=dev)
dq = torch.empty_like(qkv)
f = lambda: ops.attention_bwd(qkv, out, do, lse, dq, ws, R, H, L, L, scale, ends)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
fl = 2.5 * 4 * 64 * sum((e - b) * e for b, e in zip([0] + ends[:-1], ends)) * R * H
print(f'lib {os.environ.get("CVAR_LIB", "default")}  R={R} L={L}: {best:.3f} ms per backward  ({fl / best / 1e9:.0f} TFLOP/s algorithmic, 5 passes)  checksum {float(dq.float().abs().mean()):.6f}')
