#!/usr/bin/env python3
"""The 256x192 tile (tile_cfg 27; round 5) against the automatic plan (0) and the forced 256x256 tile (2) on the d24 shapes at the row counts of B = 8 / B = 32
generations.  Run it with CVAR_LIB=ab/libcvar_not192.so (gemm.hip built with -DCVAR_GEMM_T192=0) to see the plan without the tile in column 0.  Every result is checked
against the first column's (same K order: bit-identical expected).  The experiment also timed 192x256 and 256x128 tiles: slower than 256x192 nearly everywhere
(profiles/r05_gemm_tile_192.txt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
for M in [int(x) for x in sys.argv[1:]] or [2048, 3200, 5408, 8192, 12800]:
    for N, K in ((4608, 1536), (6144, 1536), (1536, 6144), (1536, 1536)):
        A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        ref, line = None, f'M={M:6d} N={N} K={K}:'
        for cfg in (0, 2, 27):
            ops.GEMM_TILE_CFG = cfg
            out = torch.empty(M, N, device=dev, dtype=T)
            ops.gemm(A, W, out, M=M, N=N, K=K); torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            same = torch.equal(out, ref)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): ops.gemm(A, W, out, M=M, N=N, K=K)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            line += f'  cfg{cfg:2d} {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:5.0f} TF{"" if same else " DIFF"}'
        print(line, flush=True)
ops.GEMM_TILE_CFG = 0
