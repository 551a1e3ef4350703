#!/usr/bin/env python3
"""Per-scale attention micro-benchmark (d24 geometry: H=24, head_dim 64, R = 2 x batch rows, KV arena rows of Lmax=1360)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd.spec import VarConfig
dev = torch.device('cuda:0'); T = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = VarConfig(depth=24); py = cfg.pyramid
R, H, L, C = 2 * B, cfg.H, py.L, cfg.C
qkv = (torch.randn(R, L, 3 * C, device=dev) * 0.5).to(T)
tot_ms = tot_fl = 0.0
for s, (b, e) in enumerate(zip(py.begin, py.end)):
    l = e - b
    out = torch.empty(R * l, C, device=dev, dtype=T)
    fn = lambda: ops.attention(qkv, out, R, H, L, b, l, 0.03125, None)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * R * H * l * e * 64
    tot_ms += ms; tot_fl += fl
    print(f'scale {s}: l={l:4d} kv={e:5d}  {ms:8.4f} ms  {fl / ms / 1e9:7.1f} TF/s', flush=True)
print(f'all scales: {tot_ms:.3f} ms per layer  {tot_fl / tot_ms / 1e9:.1f} TF/s')
