#!/usr/bin/env python3
"""A/B of the round-6 RPF epilogue (gate + fp32 residual read-modify-write with the residual rows prefetched by LDS DMA) against the register form (tile_cfg 28) on the
proj / fc2 shapes of d24 at the scales of a B = 512 generation, interleaved repetitions on one box.  Usage: rpf_ab.py [iters=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = 1536
for l in (512, 200, 72, 18):
    M = 1024 * l
    ada = torch.randn(1024, 6 * C, device=dev) * 0.1
    for name, N, K in (('proj', C, C), ('fc2', C, 4 * C)):
        A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        x = torch.randn(M, N, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        res = {}
        for rep in range(3):
            for cfg in (2, 28):
                ops.GEMM_TILE_CFG = cfg
                ops.gemm(A, W, x, M=M, N=N, K=K, bias=b, gate=ada, ldg=6 * C, gate_rows=l, residual=x); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): ops.gemm(A, W, x, M=M, N=N, K=K, bias=b, gate=ada, ldg=6 * C, gate_rows=l, residual=x)
                e1.record(); torch.cuda.synchronize()
                res[cfg] = min(res.get(cfg, 1e9), e0.elapsed_time(e1) / iters)
        ops.GEMM_TILE_CFG = 0
        tf = lambda ms: 2.0 * M * N * K / ms / 1e9
        print(f'{name} M={M} (l={l}) K={K}: RPF {res[2]:.3f} ms {tf(res[2]):.0f} TFLOP/s | register form {res[28]:.3f} ms {tf(res[28]):.0f} TFLOP/s | {100 * (res[28] / res[2] - 1):+.1f} %', flush=True)
        del A, W, x
