#!/usr/bin/env python3
"""adaLN LayerNorm-modulate pass at the d24 shapes: time and effective HBM bandwidth (4 B read + 2 B written per element)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0')
C = 1536; R = 768; n_ada = 6 * C * 24 + 2 * C
ada = torch.randn(R, n_ada, device=dev) * 0.1
for l in (1, 16, 64, 169, 256):
    M = R * l
    x = torch.randn(M, C, device=dev)
    u = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.ln_modulate(x, ada, 2 * C, 4 * C, n_ada, l, u, M, C, 1e-6)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'rows {M:7d} (l={l:3d}): {ms * 1e3:8.1f} us  {6.0 * M * C / ms / 1e9:6.2f} TB/s', flush=True)
