#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound row / column kernels of the training step at the d24 B=32 geometry (M = 32 x 680 rows, C = 1536):
ln_modulate, ln_modulate_bwd, gated_grad.  Buffers rotate over NSET copies (> the 256 MB infinity cache) so every call streams from HBM.
Prints ms per call and the effective rate over the ALGORITHMIC bytes (each operand read / written once).  CVAR_LIB selects an A/B build.
Usage: elem_bench.py [B=32] [depth_geometry=24] [iters=20]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops, _lib
from controlvar_amd.spec import VarConfig
dev = torch.device('cuda:0'); T = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 24
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = VarConfig(depth=depth)
C, L = cfg.C, cfg.pyramid.L
M = B * L
n_ada = 6 * C
NSET = 4
f32 = dict(device=dev, dtype=torch.float32)
X = [torch.randn(M, C, **f32) for _ in range(NSET)]
DX = [torch.randn(M, C, **f32) for _ in range(NSET)]
DY = [torch.randn(M, C, **f32).to(T) for _ in range(NSET)]
OUT16 = [torch.empty(M, C, device=dev, dtype=T) for _ in range(NSET)]
OUT32 = [torch.empty(M, C, **f32) for _ in range(NSET)]
ada = torch.randn(B, n_ada, **f32) * 0.1
dada = torch.zeros(B, n_ada, **f32)
lib = _lib.load()
nws = int(lib.cvar_train_ws_floats(M, B, C)) if hasattr(lib, 'cvar_train_ws_floats') else 2 * M + 16 * B * C
ws = torch.empty(max(nws, 2 * M + 128 * B * C), **f32)


def timed(fn):
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i % NSET)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cases = {
    'ln_modulate      (x f32 -> bf16)': (lambda i: ops.ln_modulate(X[i], ada, 2 * C, 4 * C, n_ada, L, OUT16[i], M, C, 1e-6), M * C * (4 + 2)),
    'ln_modulate_bwd  (x, dy, dx_in -> dx_out)': (lambda i: ops.ln_modulate_bwd(X[i], DY[i], ada, 2 * C, n_ada, L, DX[i], OUT32[i], dada, 2 * C, 4 * C, n_ada, M, C, 1e-6, ws),
                                                   M * C * (4 + 2 + 4 + 4)),
    'ln_modulate_bwd  in place (dx_in = dx_out)': (lambda i: ops.ln_modulate_bwd(X[i], DY[i], ada, 2 * C, n_ada, L, DX[i], DX[i], dada, 2 * C, 4 * C, n_ada, M, C, 1e-6, ws),
                                                    M * C * (4 + 2 + 4 + 4)),
    'gated_grad       (dx, f -> df)': (lambda i: ops.gated_grad(DX[i], DY[i], ada, 0, n_ada, None, OUT16[i], dada, 0, n_ada, B, L, C, ws), M * C * (4 + 2 + 2)),
}
print(f'lib {os.environ.get("CVAR_LIB", "default")}  M={M} C={C} B={B}')
for name, (fn, nbytes) in cases.items():
    best = min(timed(fn) for _ in range(3))
    print(f'{name:46s} {best * 1e3:8.1f} us  {nbytes / best / 1e9:6.2f} TB/s  ({nbytes / 1e6:.0f} MB)', flush=True)
