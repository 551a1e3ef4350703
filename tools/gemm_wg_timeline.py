#!/usr/bin/env python3
"""Workgroup timeline of one cvar_gemm launch from a -DCVAR_GEMM_TIMING build (CVAR_LIB=ab/libcvar_timing.so): per workgroup the 100 MHz
s_memrealtime at entry / first MFMA / end of the K loop / exit and the CU it ran on (HW_ID, XCC_ID).  Prints where a CU's time goes between
the tiles: dispatch gap (previous workgroup's exit -> next one's entry on the same CU), prologue (entry -> K loop), K loop, epilogue.
usage: gemm_wg_timeline.py [N=4608] [K=1536] [M=131072]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from controlvar_amd import ops, _lib
dev = torch.device('cuda:0'); T = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
M = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
ops.GEMM_TILE_CFG = int(os.environ.get('ISO_CFG', '0'))
A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
if os.environ.get('ISO_DATA') == 'zeros': A.zero_(); W.zero_()
out = torch.empty(M, N, device=dev, dtype=T)
lib = _lib.load()
for _ in range(3): ops.gemm(A, W, out, M=M, N=N, K=K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm(A, W, out, M=M, N=N, K=K); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
ntile = ((M + 255) // 256) * ((N + 255) // 256)
n = min(ntile, 16384)
buf = (ctypes.c_ulonglong * (5 * n))()
lib.cvar_gemm_dbg_wg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cvar_gemm_dbg_wg_read(buf, n) == 0
a = np.array(buf, dtype=np.uint64).reshape(n, 5)
t = a[:, :4].astype(np.int64); t0 = t[:, 0].min()
t = (t - t0) / 100.0                      # microseconds (100 MHz)
hw = a[:, 4]
cu_key = (hw & 0xffffffff) & ~np.uint64(0x3f) | ((hw >> np.uint64(32)) << np.uint64(40))     # drop wave / simd id bits [5:0], keep cu / sh / se + xcc
span = t[:, 3].max()
print(f'M={M} N={N} K={K}: {ntile} tiles (timeline of the first {n}), launch {ms * 1e3:.0f} us by events, {span:.0f} us first entry -> last exit; {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s')
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f'per workgroup (us, mean / p10 / p90): prologue {pro.mean():.2f} / {np.percentile(pro, 10):.2f} / {np.percentile(pro, 90):.2f}   K loop {loop.mean():.2f} / {np.percentile(loop, 10):.2f} / {np.percentile(loop, 90):.2f}   '
      f'epilogue {epi.mean():.2f} / {np.percentile(epi, 10):.2f} / {np.percentile(epi, 90):.2f}')
gaps, per_cu = [], {}
for i in np.argsort(t[:, 0]):
    per_cu.setdefault(int(cu_key[i]), []).append(i)
for k, idx in per_cu.items():
    for p, q in zip(idx[:-1], idx[1:]):
        gaps.append(t[q, 0] - t[p, 3])
gaps = np.array(gaps)
print(f'{len(per_cu)} distinct CUs, {np.mean([len(v) for v in per_cu.values()]):.1f} workgroups per CU; dispatch gap (exit -> next entry on the same CU): mean {gaps.mean():.2f} us, p10 {np.percentile(gaps, 10):.2f}, p50 {np.percentile(gaps, 50):.2f}, p90 {np.percentile(gaps, 90):.2f}, negative (overlap) {np.mean(gaps < 0) * 100:.0f} %')
tot = pro.mean() + loop.mean() + epi.mean() + max(gaps.mean(), 0)
print(f'share of the time of a CU: gap {max(gaps.mean(), 0) / tot * 100:.1f} %  prologue {pro.mean() / tot * 100:.1f} %  K loop {loop.mean() / tot * 100:.1f} %  epilogue {epi.mean() / tot * 100:.1f} %')
first = np.array([t[v[0], 0] for v in per_cu.values()]); last = np.array([t[v[-1], 3] for v in per_cu.values()])
print(f'ramp: first entries spread over {first.max():.1f} us; last exits between {last.min():.1f} and {last.max():.1f} us (tail {last.max() - last.min():.1f} us)')
