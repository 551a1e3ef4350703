#!/usr/bin/env python3
"""Is the GEMM main loop limited by the memory system or by its own structure?  Same FLOPs, operands aliased
through batch strides of 0 (everything L2-resident) vs the real streaming shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
def bench(fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9
for (N, K) in ((4608, 1536), (1536, 6144), (1536, 1536)):
    M = 65536
    A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
    out = torch.empty(M, N, device=dev, dtype=T)
    ms, tf = bench(lambda: ops.gemm(A, W, out, M=M, N=N, K=K), 2.0 * M * N * K)
    print(f'stream   N={N} K={K}: {ms:.3f} ms {tf:.0f} TF/s')
    nb = M // 128
    ms, tf = bench(lambda: ops.gemm(A, W, out, M=128, N=N, K=K, batch=nb, strideA=0, strideW=0, strideC=128 * N), 2.0 * M * N * K)
    print(f'aliasedA N={N} K={K}: {ms:.3f} ms {tf:.0f} TF/s')
    Az = torch.zeros_like(A); Wz = torch.zeros_like(W)
    ms, tf = bench(lambda: ops.gemm(Az, Wz, out, M=M, N=N, K=K), 2.0 * M * N * K)
    print(f'zeros    N={N} K={K}: {ms:.3f} ms {tf:.0f} TF/s')
