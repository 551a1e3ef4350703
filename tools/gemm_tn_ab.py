#!/usr/bin/env python3
"""cvar_gemm_tn (dW = dY^T X read token-major through the LDS transpose-read) against the round-1 path (two HBM transposes + cvar_gemm):
agreement and time on the d24 weight-gradient shapes (T = 32 x 1360 tokens)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T_ = torch.bfloat16


def timed(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# small exactness check against torch fp32 on the bf16-rounded operands
g = torch.Generator().manual_seed(0)
T, Nn, Kk = 96, 128, 256
A = torch.randn(T, Nn, generator=g).to(T_); B = torch.randn(T, Kk, generator=g).to(T_)
out = torch.full((Nn, Kk), float('nan'), device=dev)
ops.gemm_tn(A.to(dev), B.to(dev), out, T=T, Nn=Nn, Kk=Kk)
ref = A.float().t() @ B.float()
print('small: max |err|', (out.cpu() - ref).abs().max().item(), 'of', ref.abs().max().item())

Tt = 32 * 1360
for (Nn, Kk, name) in [(1536, 6144, 'fc2'), (6144, 1536, 'fc1'), (1536, 1536, 'proj'), (4608, 1536, 'qkv')]:
    A = torch.randn(Tt, Nn, device=dev).to(T_); B = torch.randn(Tt, Kk, device=dev).to(T_)
    Mp = Tt
    TA = torch.zeros(Nn, Mp, device=dev, dtype=T_); TB = torch.zeros(Kk, Mp, device=dev, dtype=T_)
    o_new = torch.empty(Nn, Kk, device=dev); o_old = torch.empty(Nn, Kk, device=dev)

    def old():
        ops.transpose(A, TA, 1, Tt, Nn, Nn, ld_out=Mp)
        ops.transpose(B, TB, 1, Tt, Kk, Kk, ld_out=Mp)
        ops.gemm(TA, TB, o_old, M=Nn, N=Kk, K=Mp)
    new = lambda: ops.gemm_tn(A, B, o_new, T=Tt, Nn=Nn, Kk=Kk)
    old(); new(); torch.cuda.synchronize()
    err = (o_new - o_old).abs().max().item() / o_old.abs().max().item()
    t_old, t_new = timed(old), timed(new)
    t_gemm = timed(lambda: ops.gemm(TA, TB, o_old, M=Nn, N=Kk, K=Mp))
    fl = 2.0 * Tt * Nn * Kk
    print(f'{name:5s} dW[{Nn}x{Kk}]: tn {t_new * 1e3:7.1f} us {fl / t_new / 1e9:6.0f} TF/s | transposes + gemm {t_old * 1e3:7.1f} us (gemm alone {t_gemm * 1e3:7.1f} us {fl / t_gemm / 1e9:6.0f} TF/s) | x{t_old / t_new:.2f} | rel diff {err:.2e}', flush=True)
