#!/usr/bin/env python3
"""does de-phasing the CUs (cvar_gemm_desc.stagger) help the fp32 gate + residual GEMMs now that the residual rows are prefetched through LDS (RPF)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
C = 1536
for l in (512, 72):
    M = 1024 * l
    ada = torch.randn(1024, 6 * C, device=dev) * 0.1
    for name, N, K in (('proj', C, C), ('fc2', C, 4 * C)):
        A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        x = torch.randn(M, N, device=dev) * 0.1; b = torch.randn(N, device=dev)
        res = {}
        for rep in range(2):
            for stg in (0, 20000, 40000, 80000, 160000):
                ops.GEMM_STAGGER = stg
                ops.gemm(A, W, x, M=M, N=N, K=K, bias=b, gate=ada, ldg=6 * C, gate_rows=l, residual=x); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm(A, W, x, M=M, N=N, K=K, bias=b, gate=ada, ldg=6 * C, gate_rows=l, residual=x)
                e1.record(); torch.cuda.synchronize()
                res[stg] = min(res.get(stg, 1e9), e0.elapsed_time(e1) / 10)
        ops.GEMM_STAGGER = 0
        print(f'{name} M={M} K={K}: ' + '  '.join(f'stagger {k}: {v:.3f} ms {2.0 * M * N * K / v / 1e9:.0f} TF' for k, v in res.items()), flush=True)
        del A, W, x
