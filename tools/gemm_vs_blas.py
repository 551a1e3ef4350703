#!/usr/bin/env python3
"""Yardstick only (not a product path): cvar_gemm against torch.matmul (hipBLASLt) on the same box, same data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops

dev = torch.device('cuda:0'); T = torch.bfloat16

def bench(fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9

C = 1536
DATA = os.environ.get('ISO_DATA', 'randn')          # zeros: same instruction streams without operand toggling -> what each kernel does when power does not limit the clock
print(f'data {DATA}')
for M in ((131072,) if DATA != 'randn' else (16384, 65536, 131072)):
    for name, N, K in (('qkv', 3 * C, C), ('proj', C, C), ('fc1', 4 * C, C), ('fc2', C, 4 * C)):
        A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        if DATA == 'zeros': A.zero_(); W.zero_()
        out = torch.empty(M, N, device=dev, dtype=T)
        ms0, tf0 = bench(lambda: ops.gemm(A, W, out, M=M, N=N, K=K), 2.0 * M * N * K)
        ms1, tf1 = bench(lambda: torch.matmul(A, W.t(), out=out), 2.0 * M * N * K)
        print(f'{name:5s} M={M:6d} N={N:5d} K={K:5d}  cvar {ms0:8.4f} ms {tf0:7.1f} TF/s | blas {ms1:8.4f} ms {tf1:7.1f} TF/s', flush=True)
