#!/usr/bin/env python3
"""Launch one big GEMM shape a few times (target for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
M, N, K = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 4608, int(sys.argv[2]) if len(sys.argv) > 2 else 1536
A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
out = torch.empty(M, N, device=dev, dtype=T)
for _ in range(4):
    ops.gemm(A, W, out, M=M, N=N, K=K)
torch.cuda.synchronize()
