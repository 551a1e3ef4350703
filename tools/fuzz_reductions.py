#!/usr/bin/env python3
"""Randomised sweep of the training-side reductions against torch (fp64 reference): cvar_colsum (any M incl. M < 64 and huge N,
strided input, accumulate), cvar_rowsum, the fused cross-entropy forward+backward, and the two per-sequence reductions of the backward -
cvar_ln_modulate_bwd (one-pass bf16 kernel: row tails, few rows per segment, in place / no incoming gradient; fp32 two-kernel form) and
cvar_gated_grad - with their workspace sized by cvar_train_ws_floats inside a NaN fence.  usage: fuzz_reductions.py [n_cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from controlvar_amd import ops

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    g = torch.Generator().manual_seed(case)
    kind = case % 5
    if kind == 0:                                   # column sums of an (M, N) window of a wider matrix
        M = rng.choice([1, 2, 3, 24, 63, 64, 65, 200, 4097, 43520])
        N = rng.choice([1, 7, 128, 1000, 4608, 9216, 100000])
        if M * N > 3e8:
            N = 1000
        lda = N + rng.choice([0, 0, 5, 64])
        dtype = rng.choice([torch.float32, torch.bfloat16])
        A = (torch.randn(M, lda, generator=g) * 2).to(dtype).to(dev)
        acc = rng.random() < 0.3
        out0 = torch.randn(N + 8, generator=g).to(dev)
        out = out0.clone()
        ws = torch.empty(64 * N + 16, device=dev)
        ops.colsum(A, lda, out, M, N, ws, accumulate=acc, out_off=3)
        ref = A[:, :N].double().sum(0) + (out0[3:3 + N].double() if acc else 0)
        err = (out[3:3 + N].double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        untouched = torch.equal(out[:3], out0[:3]) and torch.equal(out[3 + N:], out0[3 + N:])
        ok = err < 1e-4 and untouched                                # fp32 accumulation over up to 43520 terms
        desc = dict(kind='colsum', M=M, N=N, lda=lda, dtype=str(dtype), acc=acc)
    elif kind == 1:                                 # row sums of a (nrows, ncols) window
        nrows, ncols = rng.choice([1, 5, 128, 4096, 4608]), rng.choice([1, 3, 64, 1000, 43520])
        lda = (ncols + 7) // 8 * 8 + rng.choice([0, 8])          # rows must be 16-byte aligned (include/cvar.h); others fail loudly
        dtype = rng.choice([torch.float32, torch.bfloat16])
        A = (torch.randn(nrows, lda, generator=g) * 2).to(dtype).to(dev)
        out0 = torch.randn(nrows + 4, generator=g).to(dev)
        out = out0.clone()
        ops.rowsum(A, lda, out, nrows, ncols, out_off=2)
        ref = A[:, :ncols].double().sum(1)
        err = (out[2:2 + nrows].double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        ok = err < 1e-4 and torch.equal(out[:2], out0[:2])          # fp32 accumulation over up to 43520 terms and torch.equal(out[2 + nrows:], out0[2 + nrows:])
        desc = dict(kind='rowsum', nrows=nrows, ncols=ncols, lda=lda, dtype=str(dtype))
    elif kind == 3:                                 # adaLN LayerNorm backward: dx (row part) + per-sequence column sums d scale, d shift
        R, l = rng.choice([1, 2, 3, 7, 32]), rng.choice([1, 2, 5, 37, 170, 680, 1360])
        C = rng.choice([64, 128, 192, 1000, 1024, 1536, 1920, 2048, 2304])
        if R * l * C > 6e7:
            R = 2
        M = R * l
        dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        x = (torch.randn(M, C, generator=g) * rng.choice([0.5, 3.0]) + rng.choice([0.0, 1.5]))
        dy = torch.randn(M, C, generator=g).to(dtype)
        ada = torch.randn(R, 6 * C, generator=g) * 0.4
        mode = rng.choice(['fresh', 'inplace', 'none'])
        dx_in = torch.randn(M, C, generator=g)
        xr = x.double().requires_grad_(True)
        sc = ada[:, 2 * C:3 * C].double().clone().requires_grad_(True); sh = ada[:, 4 * C:5 * C].double().clone().requires_grad_(True)
        y = F.layer_norm(xr, (C,), eps=1e-6) * (1 + sc.repeat_interleave(l, 0)) + sh.repeat_interleave(l, 0)
        y.backward(dy.double())
        nws = ops.train_ws_floats(M, R, C)
        wsb = torch.full((nws + 512,), float('nan'), device=dev)
        ws = wsb[256:256 + nws]
        dada = torch.zeros(R, 6 * C, device=dev)
        xd, dyd, adad = x.to(dev), dy.to(dev), ada.to(dev)
        if mode == 'inplace':
            out = dx_in.to(dev).clone(); ops.ln_modulate_bwd(xd, dyd, adad, 2 * C, 6 * C, l, out, out, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
            ref_dx = dx_in.double() + xr.grad
        elif mode == 'fresh':
            out = torch.full((M, C), float('nan'), device=dev); ops.ln_modulate_bwd(xd, dyd, adad, 2 * C, 6 * C, l, dx_in.to(dev), out, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
            ref_dx = dx_in.double() + xr.grad
        else:
            out = torch.full((M, C), float('nan'), device=dev); ops.ln_modulate_bwd(xd, dyd, adad, 2 * C, 6 * C, l, None, out, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
            ref_dx = xr.grad
        e1 = ((out.double().cpu() - ref_dx).abs() / (ref_dx.abs() + 1)).max().item()
        e2 = ((dada[:, 3 * C:4 * C].double().cpu() - sc.grad).abs() / (sc.grad.abs() + l ** 0.5)).max().item()
        e3 = ((dada[:, 5 * C:].double().cpu() - sh.grad).abs() / (sh.grad.abs() + l ** 0.5)).max().item()
        fence = bool(torch.isnan(wsb[:256]).all() and torch.isnan(wsb[256 + nws:]).all()) and float(dada[:, :3 * C].abs().max()) == 0 and float(dada[:, 4 * C:5 * C].abs().max()) == 0
        err = max(e1, e2, e3)
        ok = err == err and err < 2e-4 and fence
        desc = dict(kind='ln_modulate_bwd', R=R, l=l, C=C, dtype=str(dtype), mode=mode, fence=fence)
    elif kind == 4:                                 # gated residual gradient: df = dx * gate * rowscale, dgate = rowscale * sum dx * f per sequence
        R, l = rng.choice([1, 2, 3, 7, 32]), rng.choice([1, 3, 50, 257, 680, 1360])
        C = rng.choice([64, 192, 1000, 1536, 1920])
        if R * l * C > 6e7:
            R = 2
        dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
        dx = torch.randn(R * l, C, generator=g); f = torch.randn(R * l, C, generator=g).to(dtype)
        ada = torch.randn(R, 4 * C, generator=g)
        rs = (torch.rand(R, generator=g) > 0.3).float() * 1.25 if rng.random() < 0.5 else None
        nws = ops.train_ws_floats(R * l, R, C)
        wsb = torch.full((nws + 512,), float('nan'), device=dev)
        ws = wsb[256:256 + nws]
        df = torch.full((R * l, C), float('nan'), device=dev).to(dtype); dgate = torch.zeros(R, 3 * C, device=dev)
        ops.gated_grad(dx.to(dev), f.to(dev), ada.to(dev), C, 4 * C, rs.to(dev) if rs is not None else None, df, dgate, 2 * C, 3 * C, R, l, C, ws)
        sc_r = (rs if rs is not None else torch.ones(R)).double()
        gfull = (ada[:, C:2 * C].double() * sc_r[:, None]).repeat_interleave(l, 0)
        ref_df = dx.double() * gfull
        ref_dg = (dx.double() * f.double()).view(R, l, C).sum(1) * sc_r[:, None]
        e1 = ((df.double().cpu() - ref_df).abs() / (ref_df.abs() + 1)).max().item()
        e2 = ((dgate[:, 2 * C:].double().cpu() - ref_dg).abs() / (ref_dg.abs() + l ** 0.5)).max().item()
        fence = bool(torch.isnan(wsb[:256]).all() and torch.isnan(wsb[256 + nws:]).all()) and float(dgate[:, :2 * C].abs().max()) == 0
        err = max(e1 / (1.0 if dtype == torch.float32 else 50.0), e2)
        ok = err == err and err < 2e-4 and fence
        desc = dict(kind='gated_grad', R=R, l=l, C=C, dtype=str(dtype), rowscale=rs is not None, fence=fence)
    else:                                           # fused CE
        M, V = rng.choice([1, 3, 100, 2720]), 4096
        logits = (torch.randn(M, V, generator=g) * rng.choice([1.0, 8.0, 40.0])).to(dev)
        tg = torch.randint(0, V, (M,), generator=g).to(torch.int32).to(dev)
        w = (torch.rand(M, generator=g) > 0.3).float().to(dev) if rng.random() < 0.5 else None
        gscale = rng.choice([1.0, 1.0 / M, 0.37])
        odt = rng.choice([torch.float32, torch.bfloat16])
        loss = torch.empty(M, device=dev); dl = torch.empty(M, V, device=dev, dtype=odt)
        ops.ce_fwd_bwd(logits, tg, w, gscale, loss, dl, M, V)
        lref = F.cross_entropy(logits.double(), tg.long(), reduction='none')
        p = logits.double().softmax(-1); p[torch.arange(M), tg.long()] -= 1
        dref = p * ((w.double() if w is not None else 1.0) * gscale).reshape(-1, 1) if w is not None else p * gscale
        e1 = (loss.double() - lref).abs().max().item() / max(1.0, lref.abs().max().item())
        e2 = (dl.double() - dref).abs().max().item() / max(1e-6, dref.abs().max().item())
        ok = e1 < 1e-5 and e2 < (1e-5 if odt == torch.float32 else 1e-2)
        err = max(e1, e2)
        desc = dict(kind='ce', M=M, gscale=gscale, weighted=w is not None, out=str(odt))
    if not ok:
        bad += 1
        print('FAIL', case, desc, 'err', err, flush=True)
print(f'{n_cases - bad}/{n_cases} cases ok')
sys.exit(1 if bad else 0)
