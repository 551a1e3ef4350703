#!/usr/bin/env python3
"""d24 generation throughput: eager launches (ctypes from Python) vs the captured HIP graph (models.ControlVAR.graphed_generator)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device('cuda:0'); T = torch.bfloat16
vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
var = models.build_control_var(vae, depth=24, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
labels = torch.arange(B, device=dev) % 1000; types = torch.arange(B, device=dev) % 4
eager = lambda s: var.autoregressive_infer_cfg(B, labels, g_seed=s, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
def timed(fn, n=3):
    fn(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i + 1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
te = timed(eager)
print(f'eager : {te * 1e3:8.1f} ms  {B / te:7.2f} images/s', flush=True)
var._arena = None; torch.cuda.empty_cache()            # the capture allocates its own K/V arena (arenas are per stream)
run = var.graphed_generator(B, cfg=4.0, top_k=900, top_p=0.96)
tg = timed(lambda s: run(labels, cond_type=types, g_seed=s))
print(f'graph : {tg * 1e3:8.1f} ms  {B / tg:7.2f} images/s')
