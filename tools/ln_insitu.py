#!/usr/bin/env python3
"""ln_modulate right after the GEMM that writes its input (proj with the fp32 gate + residual epilogue), as in a block, vs in isolation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
C = 1536; R = 768; l = 256; M = R * l; n_ada = 6 * C * 24 + 2 * C
ada = torch.randn(R, n_ada, device=dev) * 0.1
x = torch.randn(M, C, device=dev)
o = torch.randn(M, C, device=dev).to(T)
w = (torch.randn(C, C, device=dev) / C ** 0.5).to(T)
bias = torch.zeros(C, device=dev)
u = torch.empty(M, C, device=dev, dtype=T)
ln = lambda: ops.ln_modulate(x, ada, 2 * C, 4 * C, n_ada, l, u, M, C, 1e-6)
proj = lambda: ops.gemm(o, w, x, M=M, N=C, K=C, bias=bias, gate=ada, gate_off=0, ldg=n_ada, gate_rows=l, residual=x)
def t(fn_pre, fn, n=10):
    tot = 0.0
    for i in range(n + 2):
        if fn_pre: fn_pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: tot += e0.elapsed_time(e1)
    return tot / n
print(f'ln alone          : {t(None, ln) * 1e3:8.1f} us')
print(f'ln right after proj: {t(proj, ln) * 1e3:8.1f} us')
print(f'proj alone        : {t(None, proj) * 1e3:8.1f} us')
print(f'proj right after ln: {t(ln, proj) * 1e3:8.1f} us')
