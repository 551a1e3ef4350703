"""Reads the s_memtime instrumentation of a -DCVAR_GEMM_TIMING build (CVAR_LIB=...): per-wave cycles in the K loop."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from controlvar_amd import ops, _lib
dev = torch.device('cuda:0'); T = torch.bfloat16
M, N, K = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 6144, int(sys.argv[2]) if len(sys.argv) > 2 else 1536
A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
out = torch.empty(M, N, device=dev, dtype=T)
lib = _lib.load()
lib.cvar_gemm_dbg_tot_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
tot = (ctypes.c_ulonglong * 8)()
for _ in range(2): ops.gemm(A, W, out, M=M, N=N, K=K)
torch.cuda.synchronize(); lib.cvar_gemm_dbg_tot_read(tot, 1)
ops.gemm(A, W, out, M=M, N=N, K=K)
torch.cuda.synchronize(); lib.cvar_gemm_dbg_tot_read(tot, 1)
t = [float(x) for x in tot]
print(f'ALL TILES ({t[2]:.0f}): loop {t[0] / t[2]:.0f}  epilogue {t[1] / t[2]:.0f} (staging writes {t[6] / t[2]:.0f})  per K tile: compute {t[5] / t[2] / (K / 64 - 1):.0f} vm {t[3] / t[2] / (K / 64):.0f} barrier {t[4] / t[2] / (K / 64):.0f}')
buf = (ctypes.c_ulonglong * (64 * 8 * 8))()
lib.cvar_gemm_dbg_read.argtypes = [ctypes.c_void_p]
assert lib.cvar_gemm_dbg_read(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(64, 8, 8).astype(np.float64)
nw = 8 if os.environ.get('CVAR_GEMM_CFG') == '1' else 4
a = a[:, :nw]
nk = a[0, 0, 5]
print(f'K tiles {nk:.0f}; per K tile (cycles of s_memtime @100MHz? or shader clock): compute {a[:, :, 0].mean() / (nk - 1):.1f}  vm-wait {a[:, :, 1].mean() / nk:.1f}  barrier {a[:, :, 2].mean() / nk:.1f}')
print(f'loop total {a[:, :, 3].mean():.0f}  epilogue {a[:, :, 4].mean():.0f}')
print('per-wave (block 0):'); print(a[0, :, :5])
