#!/usr/bin/env python3
"""Does a decoder level run faster when the images go through it in sub-batches small enough for the tensors to stay in the 256 MB Infinity Cache?
One ResnetBlock of the top level (160 -> 160 at 256^2: GN, conv, GN, conv + residual) and one of the 128^2 level (160 -> 160), TOTAL images fixed, chunk varied.
Eager launches and one HIP graph per chunk (host cost out of the picture).  usage: resblock_chunk_bench.py [total images = 64]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
dev = torch.device('cuda:0')
total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
vae = models.build_vae(ch=160).to(dev)
vae._pack()
for (name, HW, C) in (('decoder.up.0.block.1', 256, 160), ('decoder.up.1.block.1', 128, 160), ('decoder.up.2.block.1', 64, 320)):
    x = torch.randn(total * HW * HW, C, device=dev).to(torch.bfloat16)
    for chunk in (1, 2, 4, 8, 16, 64):
        if chunk > total: continue
        def run():
            for s in range(0, total, chunk):
                vae._resblock(x[s * HW * HW:(s + chunk) * HW * HW], name, chunk, HW, HW, C, C)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / 2
        # graph of ONE chunk, replayed over the chunks
        xs = x[:chunk * HW * HW]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            vae._resblock(xs, name, chunk, HW, HW, C, C)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(2 * (total // chunk)): g.replay()
        e1.record(); torch.cuda.synchronize()
        gr = e0.elapsed_time(e1) / 2
        print(f'{name} {HW}^2 x{C}: {total} images in chunks of {chunk:3d}: eager {eager:8.3f} ms  graph {gr:8.3f} ms', flush=True)
