#!/usr/bin/env python3
"""GEMM / conv micro-benchmark over the shapes of the d24 hot path (run on the GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd._lib import ACT_GELU_TANH

dev = torch.device('cuda:0')
T = torch.bfloat16

def bench(fn, flops, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9

def main():
    C = 1536
    rows = []
    for M in (256, 4096, 16384, 65536):
        for name, N, K, kind in (('qkv', 3*C, C, 'remap'), ('proj', C, C, 'gate'), ('fc1', 4*C, C, 'gelu'), ('fc2', C, 4*C, 'gate'), ('head', 4096, C, 'f32out')):
            A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K**0.5).to(T)
            bias = torch.randn(N, device=dev)
            if kind == 'gate':
                x = torch.randn(M, N, device=dev); gate = torch.randn(M // 128 if M >= 128 else 1, N, device=dev)
                fn = lambda: ops.gemm(A, W, x, M=M, N=N, K=K, bias=bias, gate=gate, ldg=N, gate_rows=128 if M >= 128 else M, residual=x)
            elif kind == 'gelu':
                out = torch.empty(M, N, device=dev, dtype=T)
                fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias, act=ACT_GELU_TANH)
            elif kind == 'f32out':
                out = torch.empty(M, N, device=dev, dtype=torch.float32)
                fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias)
            else:
                out = torch.empty(M, N, device=dev, dtype=T)
                fn = lambda: ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias, remap=(M, M, 0))
            ms, tf = bench(fn, 2.0 * M * N * K)
            rows.append((name, M, N, K, round(ms, 4), round(tf, 1)))
            print(f'{name:5s} M={M:6d} N={N:5d} K={K:5d}  {ms:8.4f} ms  {tf:7.1f} TF/s', flush=True)
    # VAE convs (B=16)
    for (H, cin, cout, up) in ((256, 160, 160, 0), (128, 320, 160, 0), (64, 320, 320, 0), (16, 640, 640, 0), (128, 160, 160, 1)):
        B = 8
        Hout = H * 2 if up else H
        x = torch.randn(B * H * H, cin, device=dev).to(T); w = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(T)
        out = torch.empty(B * Hout * Hout, cout, device=dev, dtype=T); bias = torch.randn(cout, device=dev)
        fn = lambda: ops.gemm(x, w, out, M=B * Hout * Hout, N=cout, K=9 * cin, bias=bias, conv=dict(Hin=H, Win=H, Cin=cin, Hout=Hout, Wout=Hout, stride=1, up=up))
        ms, tf = bench(fn, 2.0 * B * Hout * Hout * cout * 9 * cin, iters=5)
        print(f'conv H={H:3d} {cin}->{cout} up={up}  {ms:8.4f} ms  {tf:7.1f} TF/s', flush=True)

if __name__ == '__main__':
    main()
