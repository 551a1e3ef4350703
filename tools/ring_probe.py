#!/usr/bin/env python3
"""Probe: 256x256 tile with a four-stage ring of 32-deep K tiles (gemm_skinny.hip, -DCVAR_RING_PROBE=1) against the shipped 256x256 kernel.  usage: CVAR_LIB=ab/libcvar_ring.so ring_probe.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops, _lib
dev = torch.device('cuda:0'); T = torch.bfloat16
lib = _lib.load()
fn = lib.cvar_gemm_ring_probe
fn.restype = C.c_int; fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]
M = 131072
for N, K in ((4608, 1536), (6144, 1536), (1536, 6144), (1536, 1536)):
    A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T); bias = torch.randn(N, device=dev)
    o0 = torch.empty(M, N, device=dev, dtype=T); o1 = torch.full((M, N), float('nan'), device=dev, dtype=T)
    st = torch.cuda.current_stream().cuda_stream
    def ring(): assert fn(A.data_ptr(), W.data_ptr(), bias.data_ptr(), o1.data_ptr(), M, N, K, st) == 0
    def base(): ops.gemm(A, W, o0, M=M, N=N, K=K, bias=bias)
    base(); ring(); torch.cuda.synchronize()
    diff = (o0.float() - o1.float()).abs().max().item()
    res = {}
    for name, f in (('shipped', base), ('ring4', ring), ('shipped', base), ('ring4', ring)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res[name] = min(res.get(name, 1e9), e0.elapsed_time(e1) / 5)
    print(f'N={N} K={K}: shipped {res["shipped"]:.3f} ms {2.0*M*N*K/res["shipped"]/1e9:.0f} TF/s | ring4 {res["ring4"]:.3f} ms {2.0*M*N*K/res["ring4"]/1e9:.0f} TF/s | max diff {diff:.4f}', flush=True)
