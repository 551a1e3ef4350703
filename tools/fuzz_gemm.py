#!/usr/bin/env python3
"""Randomised parity sweep of cvar_gemm (plain + conv, all epilogue combinations, awkward shapes) against torch fp32 math on
the bf16-rounded operands.  Operands live inside NaN-filled arenas, so any read outside a tensor shows up as a NaN.
usage: fuzz_gemm.py [n_cases] [seed]"""
import sys, os, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from controlvar_amd import ops
from controlvar_amd._lib import ACT_GELU_TANH, ACT_NONE

dev = torch.device('cuda:0')
ops.ensure_splitk_workspace(dev)            # split-K paths (small-M, long-K) are part of the sweep
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def arena(t, dtype):
    """copy t (cpu fp32) into the middle of a NaN-filled device buffer of `dtype`; returns the view"""
    pad = 4096
    buf = torch.full((t.numel() + 2 * pad,), float('nan'), device=dev, dtype=dtype)
    v = buf[pad:pad + t.numel()].view(t.shape)
    v.copy_(t.to(dtype))
    return v


def one_tn(case):
    """cvar_gemm_tn: dW = A^T B on token-major operands (column windows of wider NaN rows), any token count"""
    g = torch.Generator().manual_seed(case)
    T = rng.choice([1, 31, 32, 33, 200, 777, 1360, 2999])
    Nn, Kk = rng.choice([128, 256, 384, 512]), rng.choice([128, 256, 384, 512])
    lda, ldb = Nn + rng.choice([0, 8, 128]), Kk + rng.choice([0, 16, 256])
    a = torch.randn(T, Nn, generator=g); b = torch.randn(T, Kk, generator=g)
    Aw = torch.full((T, lda), float('nan')); Aw[:, :Nn] = a
    Bw = torch.full((T, ldb), float('nan')); Bw[:, :Kk] = b
    A, B = arena(Aw, torch.bfloat16), arena(Bw, torch.bfloat16)
    out = arena(torch.zeros(Nn, Kk), torch.float32)
    with_cs = rng.random() < 0.5
    cs = arena(torch.zeros(Nn), torch.float32) if with_cs else None
    ops.gemm_tn(A, B, out, T=T, Nn=Nn, Kk=Kk, lda=lda, ldb=ldb, colsum=cs)
    ref = a.to(torch.bfloat16).double().t() @ b.to(torch.bfloat16).double()
    got = out.double().cpu()
    err = ((got - ref).abs() / (ref.abs() + math.sqrt(T))).max().item() if torch.isfinite(got).all() else float('nan')
    if with_cs:                                           # bias gradient from the same pass (ones-operand MFMAs)
        cref = a.to(torch.bfloat16).double().sum(0)
        cgot = cs.double().cpu()
        e2 = ((cgot - cref).abs() / (cref.abs() + math.sqrt(T))).max().item() if torch.isfinite(cgot).all() else float('nan')
        err = max(err, e2) if e2 == e2 else float('nan')
    return err == err and err < 1e-4, f'gemm_tn T={T} {Nn}x{Kk} lda={lda} ldb={ldb} colsum={with_cs}: err {err:.3e}'


def one(case):
    if rng.random() < 0.1:
        return one_tn(case)
    dtype = torch.bfloat16 if rng.random() < 0.8 else torch.float32
    g = torch.Generator().manual_seed(case)
    conv = rng.random() < 0.4
    if conv:
        cin = rng.choice([32, 64, 96, 160, 320, 24, 8, 8])          # 8: the image conv (conv_c8.hip takes it when Cout % 160 == 0 on 16-multiple images, plain bf16 epilogue)
        cout = rng.choice([3, 16, 32, 160, 160, 320, 128, 200])
        H, W = rng.choice([(16, 16), (48, 48), (64, 40), (33, 47), (96, 96), (32, 80), (16, 48)])
        B = rng.choice([1, 2, 3])
        mode = rng.choice(['s1', 's1', 'up', 's2'])
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
        xr, wr = x.to(dtype).float(), w.to(dtype).float()
        if mode == 's1':
            ref = F.conv2d(xr, wr, None, padding=1); Ho, Wo = H, W
        elif mode == 'up':
            ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, None, padding=1); Ho, Wo = 2 * H, 2 * W
        else:
            Ho, Wo = H // 2, W // 2
            ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, None, stride=2, padding=0)[:, :, :Ho, :Wo]
        M, N, K = B * Ho * Wo, cout, 9 * cin
        A = arena(x.permute(0, 2, 3, 1).reshape(B * H * W, cin), dtype)
        Wd = arena(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin), dtype)
        ref = ref.permute(0, 2, 3, 1).reshape(M, N)
        kw = dict(conv=dict(Hin=H, Win=W, Cin=cin, Hout=Ho, Wout=Wo, stride=2 if mode == 's2' else 1, up=1 if mode == 'up' else 0))
        desc = f'conv {mode} B{B} {H}x{W} {cin}->{cout}'
    else:
        M = rng.choice([1, 7, 33, 64, 100, 200, 400, 512, 1000, 2048, 4096, 5000, 8192 + 128])
        N = rng.choice([8, 24, 100, 128, 256, 320, 512, 1536, 1920, 5760, 4096 + 256, 48, 1552, 192, 384 + 8])
        K = rng.choice([8, 40, 64, 128, 200, 512, 1536, 2048, 8192 + 64, 96, 1920, 6144])
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K)
        ref = a.to(dtype).float() @ w.to(dtype).float().t()
        A, Wd = arena(a, dtype), arena(w, dtype)
        kw = {}
        desc = f'gemm {M}x{N}x{K}'
    bias = torch.randn(N, generator=g) if rng.random() < 0.7 else None
    act = ACT_GELU_TANH if rng.random() < 0.3 else ACT_NONE
    use_gate = (not conv) and rng.random() < 0.3
    use_res = rng.random() < 0.4
    out_dtype = rng.choice([dtype, torch.float32])
    # round 6: a tenth of the big bf16 GEMMs as proj / fc2 issue them (gate + fp32 residual + fp32 output, no activation) on the forced 256x256 tile: full tiles take the
    # LDS-prefetched read-modify-write epilogue (RPF), ragged last tiles the register form, in one launch
    rpf = (not conv) and dtype == torch.bfloat16 and M >= 2048 and N >= 256 and rng.random() < 0.35
    if rpf:
        act, use_gate, use_res, out_dtype = ACT_NONE, True, True, torch.float32
    y = ref.clone()
    if bias is not None: y = y + bias
    if act == ACT_GELU_TANH: y = F.gelu(y, approximate='tanh')
    gate_rows = rng.choice([1, 3, 64, 500]) if use_gate else 1
    if use_gate:
        gt = torch.randn((M + gate_rows - 1) // gate_rows, N, generator=g)
        y = y * gt.repeat_interleave(gate_rows, 0)[:M]
    res_dtype = torch.float32 if rpf else rng.choice([dtype, torch.float32])
    if use_res:
        r = torch.randn(M, N, generator=g)
        y = y + r.to(res_dtype).float()
    # KV-arena style output row remap (qkv GEMM): row m -> (m // l) * L + off + m % l
    remap = None
    if (not conv) and (not use_res) and (not rpf) and rng.random() < 0.3:
        l = rng.choice([1, 2, 8, 50, 128, 200])
        L = l + rng.choice([0, 7, 100]); off = rng.randint(0, L - l)
        remap = (l, L, off)
    rows_out = ((M + remap[0] - 1) // remap[0]) * remap[1] if remap else M
    out = arena(torch.zeros(rows_out, N), out_dtype)
    # eligible 3x3 convs: force the LDS-halo kernel (tile_cfg 6) on most of them, whatever the grid size; the rest stay on the implicit GEMM
    ops.GEMM_TILE_CFG = 6 if conv and rng.random() < 0.8 else 0
    # plain bf16 GEMMs: a quarter of them forced onto the 256x192 tile (tile_cfg 27; in the automatic plan it only takes launches with badly filled last rounds)
    t192 = (not conv) and (not rpf) and dtype == torch.bfloat16 and rng.random() < 0.25
    if t192: ops.GEMM_TILE_CFG = 27; desc += ' [tile_cfg 27]'
    if rpf: ops.GEMM_TILE_CFG = 2; desc += ' [rpf: tile_cfg 2]'
    # plain GEMMs: half of them as the transformer issues its passes (tile_cfg 12: streaming small-M kernel / three-stage tiles where their plan applies)
    small = (not conv) and (not t192) and (not rpf) and rng.random() < 0.5
    if small: desc += ' [small_m]'
    ops.gemm(A, Wd, out, M=M, N=N, K=K, bias=arena(bias, torch.float32) if bias is not None else None, act=act,
             gate=arena(gt, torch.float32) if use_gate else None, ldg=N if use_gate else 0, gate_rows=gate_rows,
             residual=arena(r, res_dtype) if use_res else None, remap=remap, small_m=small, **kw)
    halo = ops.GEMM_TILE_CFG == 6
    ops.GEMM_TILE_CFG = 0
    if halo: desc += ' [tile_cfg 6]'
    got = out.float().cpu()
    if remap:
        l, L, off = remap
        m_idx = torch.arange(M)
        rows = (m_idx // l) * L + off + m_idx % l
        untouched = torch.ones(rows_out, dtype=torch.bool); untouched[rows] = False
        if not bool((got[untouched] == 0).all()):
            return False, f'{desc} remap {remap}: rows outside the remap were written'
        got = got[rows]
    tol = (3e-2 if out_dtype == torch.bfloat16 or dtype == torch.bfloat16 else 2e-4)
    err = ((got - y).abs() / (y.abs() + 1)).max().item() if torch.isfinite(got).all() else float('nan')
    ok = err == err and err < tol
    return ok, f'{desc} {str(dtype)[6:]}->{str(out_dtype)[6:]} bias={bias is not None} act={act} gate={gate_rows if use_gate else 0} remap={remap} res={str(res_dtype)[6:] if use_res else 0}: err {err:.3e}'


bad = 0
for c in range(n_cases):
    ok, msg = one(c)
    if not ok:
        bad += 1
        print('FAIL', c, msg, flush=True)
print(f'{n_cases - bad}/{n_cases} cases ok')
sys.exit(1 if bad else 0)
