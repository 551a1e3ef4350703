"""s_memtime instrumentation of the implicit-GEMM conv K loop: build gemm_conv.hip with -DCVAR_GEMM_TIMING (the other units without) into
a library, point CVAR_LIB at it.  usage: conv_timing.py [cin cout HW batch res]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops, _lib
dev = torch.device('cuda:0'); T = torch.bfloat16
cin, cout, HW, B, res = (int(v) for v in (sys.argv[1:6] + ['160', '160', '256', '64', '0'][len(sys.argv) - 1:]))
x = torch.randn(B * HW * HW, cin, device=dev).to(T)
w = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(T)
bias = torch.zeros(cout, device=dev)
out = torch.empty(B * HW * HW, cout, device=dev, dtype=T)
r = torch.randn(B * HW * HW, cout, device=dev).to(T) if res else None
conv = dict(Hin=HW, Win=HW, Cin=cin, Hout=HW, Wout=HW)
M, K = B * HW * HW, 9 * cin
run = lambda: ops.gemm(x, w, out, M=M, N=cout, K=K, bias=bias, conv=conv, residual=r)
lib = _lib.load()
lib.cvar_gemm_dbg_tot_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
tot = (ctypes.c_ulonglong * 8)()
for _ in range(2): run()
torch.cuda.synchronize(); lib.cvar_gemm_dbg_tot_read(tot, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize(); lib.cvar_gemm_dbg_tot_read(tot, 1)
t = [float(v) for v in tot]
nk = (K + 63) // 64
ms = e0.elapsed_time(e1)
print(f'conv {cin}->{cout} {HW}x{HW} B={B} res={res}: {ms:.3f} ms (instrumented) = {2.0 * M * cout * K / ms / 1e9:.0f} TFLOP/s; tiles {t[2]:.0f}')
print(f'per tile: loop {t[0] / t[2]:.0f}  epilogue {t[1] / t[2]:.0f} (staging writes {t[6] / t[2]:.0f}) cycles;  per K tile ({nk}): compute {t[5] / t[2] / max(nk - 1, 1):.0f}  vm-wait {t[3] / t[2] / nk:.0f}  barrier {t[4] / t[2] / nk:.0f}')
print(f'MFMA floor per K tile at 4069 flop/cycle/CU: {2.0 * 256 * cout * 64 / 4069:.0f} cycles (256-row tile)')
