#!/usr/bin/env python3
"""GroupNorm(+SiLU) pass at the VQVAE decoder's shapes: time and effective HBM bandwidth (stats: 2 B/element read; apply: 2 B read + 2 B written)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
for (B, HW, C) in [(64, 65536, 160), (64, 16384, 160), (64, 16384, 320), (64, 4096, 320), (64, 1024, 640), (64, 256, 640)]:
    x = torch.randn(B * HW, C, device=dev).to(T)
    w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=dev, dtype=torch.uint8)
    f = lambda: ops.groupnorm_silu(x, w, b, out, B, HW, C, 32, 1e-6, True, ws)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = B * HW * C
    print(f'B={B} HW={HW} C={C}: {ms * 1e3:8.1f} us  {6.0 * n / ms / 1e9:6.2f} TB/s (6 B/element over stats + apply)', flush=True)
