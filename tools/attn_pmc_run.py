#!/usr/bin/env python3
"""One attention variant at one scale of the d24 geometry, for a rocprofv3 --pmc pass: attn_pmc_run.py <v1|v2|v2p> [scale=9] [batch=128] [calls=3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd.spec import VarConfig
dev = torch.device('cuda:0'); T = torch.bfloat16
name = sys.argv[1] if len(sys.argv) > 1 else 'v2p'
si = int(sys.argv[2]) if len(sys.argv) > 2 else 9
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cfg = VarConfig(depth=24); py = cfg.pyramid
R, H, L, C = 2 * B, cfg.H, py.L, cfg.C
kv = (torch.randn(R, L, 2 * C, device=dev) * 0.5).to(T)
b, e = py.begin[si], py.end[si]; l = e - b
q = (torch.randn(R * l, C, device=dev) * 0.5).to(T)
if name == 'v2p':
    q = (q.float() * (0.03125 * 1.4426950408889634)).to(T)
out = torch.empty(R * l, C, device=dev, dtype=T)
kw = dict(v1=dict(v1=True), v2=dict(), v2p=dict(prescaled=True))[name]
for _ in range(calls): ops.attention(kv, out, R, H, L, b, l, 0.03125, None, q=q, **kw)
torch.cuda.synchronize()
