#!/usr/bin/env python3
"""Scheduling-group height (cvar_gemm_desc.group_m) sweep on the d24 shapes at the last scale's row count: time per launch.  The same
command under `rocprofv3 --pmc FETCH_SIZE` gives the fabric traffic per launch and group height (one launch per (shape, GM) when iters = 1).
usage: gemm_group_sweep.py [M=196608] [iters=10] [gms=1,2,4,8,16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gms = [int(g) for g in (sys.argv[3] if len(sys.argv) > 3 else '1,2,4,8,16').split(',')]
for name, N, K in (('qkv', 4608, 1536), ('fc1', 6144, 1536), ('fc2', 1536, 6144), ('proj', 1536, 1536)):
    A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
    out = torch.empty(M, N, device=dev, dtype=T)
    row = []
    for gm in gms:
        ops.GEMM_GROUP_M = gm
        best = 1e9
        for rep in range(3 if iters > 1 else 1):
            ops.gemm(A, W, out, M=M, N=N, K=K); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.gemm(A, W, out, M=M, N=N, K=K)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        row.append(f'GM={gm}: {best:.3f} ms {2.0 * M * N * K / best / 1e9:.0f} TF')
    ops.GEMM_GROUP_M = 0
    print(f'{name:5s} M={M} N={N} K={K}  ' + '   '.join(row), flush=True)
