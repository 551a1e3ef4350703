#!/usr/bin/env python3
"""Isolated timing of the d24 GEMM shapes (plain bf16 epilogue; no data-dependent consumers, so timing-only probe builds are safe here).
Usage: gemm_iso.py [M=131072] [iters=10]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print(f'lib {os.environ.get("CVAR_LIB", "default")}  data {os.environ.get("ISO_DATA", "randn")}  tile_cfg {os.environ.get("ISO_CFG", "0")}')
ops.GEMM_TILE_CFG = int(os.environ.get('ISO_CFG', '0'))      # 0 automatic, 2 8-wave 256x256, 3 4-wave 256x256 (128x128 per wave)
ops.GEMM_GROUP_M = int(os.environ.get('ISO_GM', '0'))          # row tiles per scheduling group (0 = automatic)
EPI = os.environ.get('ISO_EPI') == '1'          # the epilogues the transformer uses: GELU (fc1), bias + gate + fp32 residual in place (fc2 / proj)
for N, K in ((4608, 1536), (6144, 1536), (1536, 6144), (1536, 1536)):
    A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
    if os.environ.get('ISO_DATA') == 'zeros': A.zero_(); W.zero_()                    # same instruction stream, no toggling operands: shows the power / clock share
    if os.environ.get('ISO_DATA') == 'ones': A.fill_(1.0); W.fill_(1.0)
    out = torch.empty(M, N, device=dev, dtype=T)
    kw, tag = {}, 'plain'
    if EPI and N == 6144:
        kw, tag = dict(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_TANH), 'bias+gelu'
    elif EPI and N == 1536:
        out = torch.randn(M, N, device=dev) * 0.1
        ada = torch.randn(M // 256, 6 * N, device=dev) * 0.1
        kw, tag = dict(bias=torch.randn(N, device=dev), gate=ada, ldg=6 * N, gate_rows=256, residual=out), 'bias+gate+fp32 residual'
    best = 1e9
    for rep in range(3):
        ops.gemm(A, W, out, M=M, N=N, K=K, **kw); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.gemm(A, W, out, M=M, N=N, K=K, **kw)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    print(f'M={M} N={N} K={K} {tag}: {best:.3f} ms  {2.0 * M * N * K / best / 1e9:.0f} TFLOP/s', flush=True)
