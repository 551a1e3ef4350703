#!/usr/bin/env python3
"""LDS-halo 3x3 conv (conv_halo.hip, the default for eligible shapes) against the implicit-GEMM tiles (ops.GEMM_TILE_CFG = 5 keeps a
conv on them): agreement and isolated timing on the VQVAE decoder's shapes.  usage: conv_halo_ab.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlvar_amd import ops

dev = torch.device('cuda:0'); T = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def run(x, w, bias, out, r, B, HW, cin, cout, cfg, up=0):
    """HW: output size; up=1: the input is (HW/2)^2 and is read through a nearest x2 upsample"""
    ops.GEMM_TILE_CFG = cfg
    hin = HW // 2 if up else HW
    ops.gemm(x, w, out, M=B * HW * HW, N=cout, K=9 * cin, bias=bias, residual=r, conv=dict(Hin=hin, Win=hin, Cin=cin, Hout=HW, Wout=HW, up=up))
    ops.GEMM_TILE_CFG = 0


def timed(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# correctness on a small case against torch fp32
g = torch.Generator().manual_seed(0)
B, HW, cin, cout = 2, 32, 64, 160
xc = torch.randn(B, cin, HW, HW, generator=g); wc = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5; bc = torch.randn(cout, generator=g)
rc = torch.randn(B, cout, HW, HW, generator=g)
x = xc.permute(0, 2, 3, 1).reshape(-1, cin).to(dev, T); w = wc.permute(0, 2, 3, 1).reshape(cout, 9 * cin).to(dev, T)
r = rc.permute(0, 2, 3, 1).reshape(-1, cout).to(dev, T)
ref = F.conv2d(x.float().cpu().view(B, HW, HW, cin).permute(0, 3, 1, 2), w.float().cpu().view(cout, 3, 3, cin).permute(0, 3, 1, 2), bc, padding=1) \
    + r.float().cpu().view(B, HW, HW, cout).permute(0, 3, 1, 2)
for cfg in (0, 5):
    out = torch.full((B * HW * HW, cout), float('nan'), device=dev, dtype=T)
    run(x, w, bc.to(dev), out, r, B, HW, cin, cout, cfg)
    got = out.float().cpu().view(B, HW, HW, cout).permute(0, 3, 1, 2)
    print(f'cfg {cfg}: max |got - torch fp32| = {(got - ref).abs().max().item():.4f} (bf16 output rounding ~ {ref.abs().max().item() / 256:.4f})')

# conv_out (160 -> 3, fp32 output): narrow form
B, HW, cin, cout = 64, 256, 160, 3
x = torch.randn(B * HW * HW, cin, device=dev).to(T); w = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(T)
bias = torch.randn(cout, device=dev)
o0 = torch.empty(B * HW * HW, cout, device=dev, dtype=torch.float32); o5 = torch.empty_like(o0)
run(x, w, bias, o0, None, B, HW, cin, cout, 0); run(x, w, bias, o5, None, B, HW, cin, cout, 5)
t0 = timed(lambda: run(x, w, bias, o0, None, B, HW, cin, cout, 0)); t5 = timed(lambda: run(x, w, bias, o5, None, B, HW, cin, cout, 5))
print(f' 160->   3 256^2 B=64 fp32 out (conv_out): halo {t0:7.3f} ms | implicit-GEMM {t5:7.3f} ms | x{t5 / t0:.3f} | max diff {(o0 - o5).abs().max().item():.2e}', flush=True)
del x, w, o0, o5

for (B, HW, cin, cout, res, up) in [(64, 256, 160, 160, 0, 0), (64, 256, 160, 160, 1, 0), (64, 128, 320, 320, 0, 0), (64, 128, 160, 160, 1, 0), (64, 128, 320, 160, 0, 0),
                                    (64, 64, 320, 320, 1, 0), (64, 32, 640, 640, 0, 0), (64, 16, 640, 640, 1, 0),
                                    (64, 256, 160, 160, 0, 1), (64, 128, 320, 320, 0, 1), (64, 64, 320, 320, 0, 1), (64, 32, 640, 640, 0, 1)]:
    hin = HW // 2 if up else HW
    x = torch.randn(B * hin * hin, cin, device=dev).to(T); w = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(T)
    bias = torch.randn(cout, device=dev); r = torch.randn(B * HW * HW, cout, device=dev).to(T) if res else None
    o0 = torch.empty(B * HW * HW, cout, device=dev, dtype=T); o5 = torch.empty_like(o0)
    run(x, w, bias, o0, r, B, HW, cin, cout, 0, up); run(x, w, bias, o5, r, B, HW, cin, cout, 5, up)
    diff = (o0.float() - o5.float()).abs().max().item()
    t0 = timed(lambda: run(x, w, bias, o0, r, B, HW, cin, cout, 0, up)); t5 = timed(lambda: run(x, w, bias, o5, r, B, HW, cin, cout, 5, up))
    fl = 2.0 * B * HW * HW * cout * 9 * cin
    print(f'{cin:4d}->{cout:4d} {HW:3d}^2 B={B} res={res} up={up}: halo {t0:7.3f} ms {fl / t0 / 1e9:6.0f} TF/s | implicit-GEMM {t5:7.3f} ms {fl / t5 / 1e9:6.0f} TF/s | x{t5 / t0:.3f} | max diff {diff:.4f}', flush=True)
