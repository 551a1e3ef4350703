#!/usr/bin/env python3
"""Small-M GEMMs of one d24 block, per call: the weight-streaming kernel (tile_cfg 12) against the LDS-tiled kernels + split-K (tile_cfg 0), each as the
model issues it (epilogue, adaLN request).  One HIP graph of 24 different-weight calls per measurement, so that launches overlap as in a generation and
the weights come from HBM, not from the Infinity Cache.  usage: skinny_bench.py [M ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd._lib import ACT_GELU_TANH

ops.GEMM_TILE_CFG = int(os.environ.get('ISO_CFG', '0'))          # tile_cfg of the LDS-tiled arm (13 / 14: three / four LDS stages on the 128x128 tile)
dev = torch.device('cuda:0')
C, depth, Lmax = 1536, int(os.environ.get('SK_DEPTH', '24')), 1360       # SK_DEPTH=2: 113 MB of weights - they stay in the 256 MB Infinity Cache between replays
Ms = [int(a) for a in sys.argv[1:]] or [4, 16, 36, 64, 100, 144, 256, 400, 676, 1024]
g = torch.Generator(device='cpu').manual_seed(0)
Wqkv = (torch.randn(depth, 3 * C, C, generator=g) * 0.02).to(torch.bfloat16).to(dev)
Wproj = (torch.randn(depth, C, C, generator=g) * 0.02).to(torch.bfloat16).to(dev)
Wfc1 = (torch.randn(depth, 4 * C, C, generator=g) * 0.02).to(torch.bfloat16).to(dev)
Wfc2 = (torch.randn(depth, C, 4 * C, generator=g) * 0.02).to(torch.bfloat16).to(dev)
bq, bp, b1, b2 = (torch.randn(depth, n, generator=g).to(dev) for n in (3 * C, C, 4 * C, C))
ops.ensure_splitk_workspace(dev)


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / depth       # us per call


for M in Ms:
    R = 2
    l = max(M // R, 1)
    Mx = R * l
    n_ada = 6 * C * depth + 2 * C
    ada = (torch.randn(R, n_ada, generator=g) * 0.2).to(dev)
    u = (torch.randn(Mx, C, generator=g)).to(torch.bfloat16).to(dev)
    h = (torch.randn(Mx, 4 * C, generator=g)).to(torch.bfloat16).to(dev)
    x = torch.randn(Mx, C, generator=g).to(dev)
    hb = torch.empty(Mx, 4 * C, device=dev, dtype=torch.bfloat16)
    uo = torch.empty(Mx, C, device=dev, dtype=torch.bfloat16)
    qs = torch.empty(Mx, C, device=dev, dtype=torch.bfloat16)
    arena = torch.empty(depth, R, Lmax, 2 * C, device=dev, dtype=torch.bfloat16)
    row = {}
    for small in (True, False):
        def qkv():
            for i in range(depth):
                ops.gemm(u, Wqkv, arena, M=Mx, N=3 * C, K=C, w_off=i * 3 * C * C, bias=bq[i], c_off=i * R * Lmax * 2 * C, ldc=2 * C, remap=(l, Lmax, 7),
                         split=(qs, C, C), split_alpha=0.18, small_m=small)
        def proj():
            for i in range(depth):
                a0 = i * 6 * C
                ops.gemm(u, Wproj, x, M=Mx, N=C, K=C, w_off=i * C * C, bias=bp[i], gate=ada, gate_off=a0, ldg=n_ada, gate_rows=l, residual=x,
                         ln=(uo, ada, a0 + 3 * C, a0 + 5 * C, n_ada, l, 1e-6), small_m=small)
        def fc1():
            for i in range(depth):
                ops.gemm(u, Wfc1, hb, M=Mx, N=4 * C, K=C, w_off=i * 4 * C * C, bias=b1[i], act=ACT_GELU_TANH, small_m=small)
        def fc2():
            for i in range(depth):
                a0 = i * 6 * C
                ops.gemm(h, Wfc2, x, M=Mx, N=C, K=4 * C, w_off=i * 4 * C * C, bias=b2[i], gate=ada, gate_off=a0 + C, ldg=n_ada, gate_rows=l, residual=x,
                         ln=(uo, ada, a0 + 2 * C, a0 + 4 * C, n_ada, l, 1e-6), small_m=small)
        row[small] = [bench(f) for f in (qkv, proj, fc1, fc2)]
    s, t = row[True], row[False]
    print('M=%5d  streaming: qkv %6.1f proj+ln %6.1f fc1 %6.1f fc2+ln %6.1f  sum %6.1f us | tiled: qkv %6.1f proj+ln %6.1f fc1 %6.1f fc2+ln %6.1f  sum %6.1f us' %
          (Mx, s[0], s[1], s[2], s[3], sum(s), t[0], t[1], t[2], t[3], sum(t)), flush=True)
