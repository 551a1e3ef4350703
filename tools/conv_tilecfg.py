#!/usr/bin/env python3
"""A/B of the 256x160 implicit-conv tile: 4 waves (one per SIMD, default) vs 8 waves (tile_cfg 4) on the VQVAE decoder's conv shapes; checks equality."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, cin, cout, res) in ((16, 256, 160, 160, False), (16, 256, 160, 160, True), (32, 128, 160, 160, True), (32, 64, 320, 320, False), (64, 16, 640, 640, True), (32, 128, 320, 160, False)):
    x = torch.randn(B * H * H, cin, device=dev).to(T); w = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(T)
    bias = torch.randn(cout, device=dev); r = torch.randn(B * H * H, cout, device=dev).to(T) if res else None
    outs = {}
    res_t = {}
    for rep in range(2):
        for arm in (0, 4):
            out = torch.empty(B * H * H, cout, device=dev, dtype=T)
            ops.GEMM_TILE_CFG = arm
            fn = lambda: ops.gemm(x, w, out, M=B * H * H, N=cout, K=9 * cin, bias=bias, residual=r, conv=dict(Hin=H, Win=H, Cin=cin, Hout=H, Wout=H, stride=1, up=0))
            ms = timeit(fn)
            res_t[arm] = min(res_t.get(arm, 1e9), ms); outs[arm] = out
    ops.GEMM_TILE_CFG = 0
    fl = 2.0 * B * H * H * cout * 9 * cin
    print(f'conv B={B} {H}x{H} {cin}->{cout} res={int(res)}: 4-wave {res_t[0]:.3f} ms {fl / res_t[0] / 1e9:6.0f} TF | 8-wave {res_t[4]:.3f} ms {fl / res_t[4] / 1e9:6.0f} TF | equal {torch.equal(outs[0], outs[4])}', flush=True)
