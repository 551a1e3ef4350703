#!/usr/bin/env python3
"""A/B of the tile choice (cvar_gemm_desc::tile_cfg: 0 auto = 8-wave 256x256, 1 = 128x128 with two workgroups per CU, 3 = 4-wave 256x256)
on the d24 GEMM shapes, isolated, interleaved arms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd._lib import ACT_GELU_TANH
dev = torch.device('cuda:0'); T = torch.bfloat16
C = 1536
ARMS = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 3]
Ms = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [131072, 12800, 4608]

def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for M in Ms:
    for name, N, K, kind in (('qkv', 3 * C, C, 'remap'), ('proj', C, C, 'gate'), ('fc1', 4 * C, C, 'gelu'), ('fc2', C, 4 * C, 'gate'), ('head', 4096, C, 'f32out')):
        A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        bias = torch.randn(N, device=dev)
        if kind == 'gate':
            x = torch.randn(M, N, device=dev); gate = torch.randn(M // 128, N, device=dev)
            fn = lambda: ops.gemm(A, W, x, M=M, N=N, K=K, bias=bias, gate=gate, ldg=N, gate_rows=128, residual=x)
        elif kind == 'gelu':
            out = torch.empty(M, N, device=dev, dtype=T)
            fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias, act=ACT_GELU_TANH)
        elif kind == 'f32out':
            out = torch.empty(M, N, device=dev, dtype=torch.float32)
            fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias)
        else:
            out = torch.empty(M, N, device=dev, dtype=T)
            fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias, remap=(M, M, 0))
        res = []
        for rep in range(2):
            for arm in ARMS:
                ops.GEMM_TILE_CFG = arm
                res.append((arm, timeit(fn)))
        ops.GEMM_TILE_CFG = 0
        best = {a: min(ms for aa, ms in res if aa == a) for a in ARMS}
        print(f'{name:5s} M={M:6d} N={N:5d} K={K:5d}  ' + '  '.join(f'cfg{a}: {ms:.3f}ms {2.0 * M * N * K / ms / 1e9:6.0f}TF' for a, ms in best.items()), flush=True)
