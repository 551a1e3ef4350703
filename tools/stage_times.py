#!/usr/bin/env python3
"""Per-scale wall time of one d24 generation (events between scales) + per-op micro timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, ops
dev = torch.device('cuda:0')
depth, B = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 64
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True).to(dev).eval()
labels = torch.arange(B, device=dev) % 1000; types = torch.arange(B, device=dev) % 4
orig = var._blocks_and_head
evs = []
def wrapped(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(*a, **k); e1.record(); evs.append((e0, e1)); return out
var._blocks_and_head = wrapped
for it in range(2):
    evs.clear()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); var.autoregressive_infer_cfg(B, labels, g_seed=1, cfg=4.0, top_k=900, top_p=0.96, cond_type=types); t1.record()
    torch.cuda.synchronize()
tot = t0.elapsed_time(t1)
st = [a.elapsed_time(b) for a, b in evs]
print('total %.1f ms; blocks+head per scale: %s ; sum %.1f ms; rest (sampler, pyramid, decode) %.1f ms' % (tot, ' '.join('%.1f' % x for x in st), sum(st), tot - sum(st)))
# ln_modulate micro
C = 64 * depth; M = 2 * B * 512
x = torch.randn(M, C, device=dev); ada = torch.randn(2 * B, 6 * C, device=dev); out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
for _ in range(3): ops.ln_modulate(x, ada, 2 * C, 4 * C, 6 * C, 512, out, M, C, 1e-6)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.ln_modulate(x, ada, 2 * C, 4 * C, 6 * C, 512, out, M, C, 1e-6)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('ln_modulate M=%d C=%d: %.3f ms  %.2f TB/s' % (M, C, ms, M * C * 6 / ms / 1e9))
