#!/usr/bin/env python3
"""In-situ GEMM table: one d24 generation with per-launch HIP events, aggregated by (kind, M, N, K, epilogue)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, ops

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0'); T = torch.bfloat16
vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
var = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
labels = torch.arange(B, device=dev) % 1000; types = torch.arange(B, device=dev) % 4
run = lambda s: var.autoregressive_infer_cfg(B, labels, g_seed=s, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
run(0); torch.cuda.synchronize()
ops.GEMM_PROFILE = prof = []
run(1); torch.cuda.synchronize()
ops.GEMM_PROFILE = None
acc = collections.OrderedDict()
for e0, e1, fl, tag in prof:
    a = acc.setdefault(tag, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in acc.values())
print(f'total GEMM ms {tot:.1f}  TF/s {sum(a[2] for a in acc.values()) / tot / 1e9:.1f}')
for tag, (n, ms, fl) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{str(tag):70s} n={n:4d} {ms:8.2f} ms {100 * ms / tot:5.1f}%  {fl / ms / 1e9:7.1f} TF/s')
