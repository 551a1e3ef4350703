#!/usr/bin/env python3
"""A/B of two consecutive output tiles per workgroup (round 6, cvar_gemm_kernel) against one (tile_cfg 29) on the d24 shapes with their real epilogues, interleaved on one
box; also checks that the results are bit-identical.  Usage: tpw_ab.py [M=524288] [iters=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C = 1536; l = 512
ada = torch.randn(max(M // l, 1), 6 * C, device=dev) * 0.1
for name, N, K, kind in (('qkv', 3 * C, C, 'plain'), ('fc1', 4 * C, C, 'gelu'), ('fc2', C, 4 * C, 'gate'), ('proj', C, C, 'gate'), ('head', 4096, C, 'f32')):
    A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T); b = torch.randn(N, device=dev)
    x0 = torch.randn(M, N, device=dev) * 0.1 if kind == 'gate' else None

    def call(out):
        if kind == 'plain': ops.gemm(A, W, out, M=M, N=N, K=K, bias=b)
        elif kind == 'gelu': ops.gemm(A, W, out, M=M, N=N, K=K, bias=b, act=ops.ACT_GELU_TANH)
        elif kind == 'f32': ops.gemm(A, W, out, M=M, N=N, K=K, bias=b)
        else: ops.gemm(A, W, out, M=M, N=N, K=K, bias=b, gate=ada, ldg=6 * C, gate_rows=l, residual=out)
    res, outs = {}, {}
    for rep in range(3):
        for cfg in (0, 29):
            ops.GEMM_TILE_CFG = cfg
            out = x0.clone() if kind == 'gate' else torch.empty(M, N, device=dev, dtype=torch.float32 if kind == 'f32' else T)
            call(out); torch.cuda.synchronize()
            if rep == 0: outs[cfg] = out.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call(out)
            e1.record(); torch.cuda.synchronize()
            res[cfg] = min(res.get(cfg, 1e9), e0.elapsed_time(e1) / iters)
            del out
    ops.GEMM_TILE_CFG = 0
    tf = lambda ms: 2.0 * M * N * K / ms / 1e9
    print(f'{name} M={M} N={N} K={K} {kind}: two tiles per workgroup {res[0]:.3f} ms {tf(res[0]):.0f} TFLOP/s | one {res[29]:.3f} ms {tf(res[29]):.0f} TFLOP/s | {100 * (res[29] / res[0] - 1):+.1f} %  bit-identical {torch.equal(outs[0], outs[29])}', flush=True)
    del A, W, outs
