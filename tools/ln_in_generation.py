#!/usr/bin/env python3
"""Per-scale time of ln_modulate inside a real d24 generation (HIP events around every call)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device('cuda:0'); T = torch.bfloat16
vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
var = models.build_control_var(vae, depth=24, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
labels = torch.arange(B, device=dev) % 1000; types = torch.arange(B, device=dev) % 4
run = lambda s: var.autoregressive_infer_cfg(B, labels, g_seed=s, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
run(0); torch.cuda.synchronize()
rec = []
orig = ops.ln_modulate
def wrapped(x, ada, so, sh, ld, rows_per, out, M, C, eps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(x, ada, so, sh, ld, rows_per, out, M, C, eps); e1.record()
    rec.append((M, rows_per, e0, e1))
    return r
ops.ln_modulate = wrapped
models.ops.ln_modulate = wrapped
run(1); torch.cuda.synchronize()
acc = collections.OrderedDict()
for M, l, e0, e1 in rec:
    a = acc.setdefault((M, l), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = 0
for (M, l), (n, ms) in acc.items():
    tot += ms
    print(f'M={M:7d} l={l:3d}: n={n:3d} avg {ms / n * 1e3:8.1f} us  {6.0 * M * 1536 / (ms / n) / 1e9:6.2f} TB/s')
print(f'total {tot:.1f} ms per generation')
