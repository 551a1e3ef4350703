#!/usr/bin/env python3
"""Throughput of d24 generation when TWO generations are in flight on two HIP streams (each its own KV arena), issued alternately by
one host thread: the small scales / tails of one overlap the large-scale GEMMs and the VQVAE decodes of the other.
usage: two_stream.py <B per generation> <generations> [streams]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device('cuda:0'); T = torch.bfloat16
vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
var = models.build_control_var(vae, depth=24, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
labels = (torch.arange(B) % 1000).to(dev); types = (torch.arange(B) % 4).to(dev)
var._pack(); vae._pack()
streams = [torch.cuda.Stream() for _ in range(NS)]
def gen(i):
    with torch.cuda.stream(streams[i % NS]):
        return var.autoregressive_infer_cfg(B, labels, g_seed=100 + i, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
for i in range(NS): gen(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = [gen(i) for i in range(G)]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'streams={NS} B={B} x {G} generations: {B * G / dt:.1f} images/s ({dt / G * 1e3:.1f} ms per generation issued)')
