"""Training-side kernels on MI355X against torch autograd (CPU fp32) of the same ops."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from controlvar_amd import ops  # noqa: E402
from controlvar_amd.spec import Pyramid  # noqa: E402

F32, BF16 = torch.float32, torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t, dtype, d):
    return t.to(dtype).to(d).contiguous()


def close(got, ref, dtype, f32_tol=1e-4, bf16_rel=2e-2):
    got, ref = got.float().cpu(), ref.float()
    tol = f32_tol if dtype == F32 else bf16_rel
    return bool(((got - ref).abs() <= tol * (ref.abs() + 1)).all())


@pytest.mark.parametrize('dtype', [F32, BF16])
def test_gate_residual_and_gated_grad(gpu_device, dtype):
    R, l, C = 3, 50, 192
    x, f = rnd(R * l, C, seed=1), rnd(R * l, C, seed=2)
    ada = rnd(R, 4 * C, seed=3)
    rs = torch.tensor([1.0, 0.0, 1.25])
    fd = dev(f, dtype, gpu_device)
    xd = x.to(gpu_device).clone()
    ops.gate_residual(xd, fd, ada.to(gpu_device), C, 4 * C, l, rs.to(gpu_device), R * l, C)
    g = (ada[:, C:2 * C] * rs[:, None]).repeat_interleave(l, 0)
    ff = fd.float().cpu()
    assert close(xd, x + g * ff, F32, 1e-5)
    dx = rnd(R * l, C, seed=4)
    df = torch.empty(R * l, C, device=gpu_device, dtype=dtype)
    dgate = torch.zeros(R, 3 * C, device=gpu_device)
    ws = torch.empty(ops.train_ws_floats(R * l, R, C), device=gpu_device)
    ops.gated_grad(dx.to(gpu_device), fd, ada.to(gpu_device), C, 4 * C, rs.to(gpu_device), df, dgate, 2 * C, 3 * C, R, l, C, ws)
    assert close(df, dx * g, dtype)
    ref = (dx * ff).view(R, l, C).sum(1) * rs[:, None]
    assert close(dgate[:, 2 * C:], ref, F32, 1e-4) and dgate[:, :2 * C].abs().max() == 0
    # ABI 16: an undersized workspace is refused instead of being written past its end
    from controlvar_amd._lib import CvarError
    with pytest.raises(CvarError):
        ops.gated_grad(dx.to(gpu_device), fd, ada.to(gpu_device), C, 4 * C, rs.to(gpu_device), df, dgate, 2 * C, 3 * C, R, l, C, ws[:ws.numel() - 1])
    x0 = x.to(gpu_device)
    with pytest.raises(CvarError):
        ops.ln_modulate_bwd(x0, fd, ada.to(gpu_device), 0, 4 * C, l, None, torch.empty_like(x0), dgate, 0, C, 3 * C, R * l, C, 1e-6, ws[:ws.numel() - 1])


@pytest.mark.parametrize('dtype', [F32, BF16])
def test_gelu_fwd_bwd(gpu_device, dtype):
    a = rnd(1000, 33, seed=1, scale=2.0)
    ad = dev(a, dtype, gpu_device)
    h = torch.empty_like(ad)
    ops.gelu(ad, h)
    ar = ad.float().cpu().requires_grad_(True)
    ref = F.gelu(ar, approximate='tanh')
    assert close(h, ref.detach(), dtype, 1e-5)
    dh = rnd(1000, 33, seed=2)
    dhd = dev(dh, dtype, gpu_device)
    ref.backward(dhd.float().cpu())
    ops.gelu_bwd(ad, dhd)
    assert close(dhd, ar.grad, dtype, 1e-5)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('C,R,l', [(128, 2, 37), (1536, 2, 37), (1000, 3, 70), (1024, 5, 3), (1536, 2, 680), (2048, 1, 9)])
def test_ln_modulate_bwd(gpu_device, dtype, C, R, l):
    """row part (dx) and the per-sequence column sums (d scale, d shift); bf16 takes the one-pass kernel (row tails C % 256 != 0, fewer rows
    than waves, many segments), fp32 the two-kernel form"""
    M = R * l
    x = (rnd(M, C, seed=1, scale=1.7) + 0.3).requires_grad_(True)
    ada = rnd(R, 6 * C, seed=2, scale=0.4)
    sc = ada[:, 2 * C:3 * C].clone().requires_grad_(True)
    sh = ada[:, 4 * C:5 * C].clone().requires_grad_(True)
    y = F.layer_norm(x, (C,), eps=1e-6) * (1 + sc.repeat_interleave(l, 0)) + sh.repeat_interleave(l, 0)
    dy = rnd(M, C, seed=3)
    dyd = dev(dy, dtype, gpu_device)
    y.backward(dyd.float().cpu())
    dx_in = rnd(M, C, seed=4)
    dx_out = torch.empty(M, C, device=gpu_device)
    dada = torch.zeros(R, 6 * C, device=gpu_device)
    ws = torch.empty(ops.train_ws_floats(M, R, C), device=gpu_device)
    ops.ln_modulate_bwd(x.detach().to(gpu_device), dyd, ada.to(gpu_device), 2 * C, 6 * C, l, dx_in.to(gpu_device), dx_out, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
    assert close(dx_out, dx_in + x.grad, F32, 2e-4)
    assert close(dada[:, 3 * C:4 * C], sc.grad, F32, 2e-4) and close(dada[:, 5 * C:], sh.grad, F32, 2e-4)
    assert dada[:, :3 * C].abs().max() == 0 and dada[:, 4 * C:5 * C].abs().max() == 0
    # in place (dx_in is dx_out: how the step accumulates the residual stream's gradient) and without an incoming gradient
    acc = dx_in.to(gpu_device).clone()
    ops.ln_modulate_bwd(x.detach().to(gpu_device), dyd, ada.to(gpu_device), 2 * C, 6 * C, l, acc, acc, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
    assert torch.equal(acc, dx_out)
    ops.ln_modulate_bwd(x.detach().to(gpu_device), dyd, ada.to(gpu_device), 2 * C, 6 * C, l, None, acc, dada, 3 * C, 5 * C, 6 * C, M, C, 1e-6, ws)
    assert close(acc, x.grad, F32, 2e-4)


@pytest.mark.parametrize('dtype', [F32, BF16])
def test_colsum_and_transpose_padded(gpu_device, dtype):
    M, N = 1000, 200
    A = rnd(M, 3 * N, seed=1)
    Ad = dev(A, dtype, gpu_device)
    out = torch.ones(N, device=gpu_device)
    ws = torch.empty(64 * N, device=gpu_device)
    ops.colsum(Ad, 3 * N, out, M, N, ws, accumulate=True, a_off=N)
    assert close(out, 1 + Ad.float().cpu()[:, N:2 * N].sum(0), F32, 1e-4)
    Mp = 1008
    T_ = torch.zeros(N, Mp, device=gpu_device, dtype=dtype)
    ops.transpose(Ad, T_, 1, M, N, 3 * N, in_off=N, ld_out=Mp)
    assert torch.equal(T_[:, :M].float().cpu(), Ad.float().cpu()[:, N:2 * N].t()) and T_[:, M:].abs().max() == 0


@pytest.mark.parametrize('out_dtype', [F32, BF16])
def test_ce_fwd_bwd(gpu_device, out_dtype):
    M, V = 300, 4096
    logits = rnd(M, V, seed=1, scale=3.0).requires_grad_(True)
    tgt = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2))
    w = torch.rand(M, generator=torch.Generator().manual_seed(3))
    lt = F.cross_entropy(logits, tgt, reduction='none')
    scale = 1.0 / (M * (w.mean().item() + 1e-6))
    (lt * w).sum().mul(scale).backward()
    loss_tok = torch.empty(M, device=gpu_device)
    dl = torch.empty(M, V, device=gpu_device, dtype=out_dtype)
    ops.ce_fwd_bwd(logits.detach().to(gpu_device), tgt.int().to(gpu_device), w.to(gpu_device), scale, loss_tok, dl, M, V)
    assert close(loss_tok, lt.detach(), F32, 1e-5)
    assert (dl.float().cpu() - logits.grad).abs().max() < (1e-8 if out_dtype == F32 else 2e-2 * logits.grad.abs().max().item())


@pytest.mark.parametrize('dtype', [F32, BF16])
def test_attention_lse_and_backward(gpu_device, dtype):
    """level-masked attention over the full ControlVAR pyramid: forward lse + dQ/dK/dV vs autograd of the masked softmax"""
    py = Pyramid()
    R, H, c, L = 1, 2, 64, py.L
    C3 = 3 * H * c
    qkv = rnd(R, L, C3, seed=5, scale=1.2)
    qd = dev(qkv, dtype, gpu_device)
    scale = 0.125
    lvl_end = list(py.end)
    out = torch.empty(R * L, H * c, device=gpu_device, dtype=dtype)
    lse = torch.empty(R, H, L, device=gpu_device)
    ops.attention(qd, out, R, H, L, 0, L, scale, lvl_end, lse=lse)
    x = qd.float().cpu().view(R, L, 3, H, c).requires_grad_(True)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    lvl = torch.from_numpy(py.level_of_token())
    bias = torch.where(lvl.view(-1, 1) >= lvl.view(1, -1), 0., -torch.inf)
    s = q @ k.transpose(-1, -2) * scale + bias
    o_ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(R * L, H * c)
    assert close(out, o_ref.detach(), dtype, 2e-5)
    assert close(lse, torch.logsumexp(s, dim=-1).detach(), F32, 1e-4 if dtype == F32 else 2e-2)
    do = rnd(R * L, H * c, seed=6)
    dod = dev(do, dtype, gpu_device)
    o_ref.backward(dod.float().cpu())
    dqkv = torch.zeros(R, L, C3, device=gpu_device, dtype=dtype)
    ws = torch.empty(R * H * L, device=gpu_device)
    # the backward consumes the forward's own (possibly bf16-rounded) output
    ops.attention_bwd(qd, out, dod, lse, dqkv, ws, R, H, L, L, scale, lvl_end)
    ref = x.grad.view(R, L, C3)
    err = (dqkv.float().cpu() - ref).abs().max().item()
    assert err < (2e-4 if dtype == F32 else 3e-2) * max(1.0, ref.abs().max().item()), err
    if dtype == BF16:       # MFMA kernels (default) against the exact row-wise kernels of the same library
        dq_rw = torch.zeros_like(dqkv)
        ops.attention_bwd(qd, out, dod, lse, dq_rw, ws, R, H, L, L, scale, lvl_end, rowwise=True)
        d = (dqkv.float() - dq_rw.float()).abs().max().item()
        assert d < 3e-2 * max(1.0, ref.abs().max().item()), d
        assert (dq_rw.float().cpu() - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


def test_adamw_sumsq_clip_scatter_silu(gpu_device):
    from oracle.train_ref import adamw_update
    n = 10007
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    m0, v0 = rnd(n, seed=3) * 0.1, rnd(n, seed=4).abs() * 0.01
    pd, md, vd = p.to(gpu_device).clone(), m0.to(gpu_device).clone(), v0.to(gpu_device).clone()
    gs = torch.tensor([0.5], device=gpu_device)
    ops.adamw(pd, g.to(gpu_device), md, vd, 3e-3, 0.9, 0.95, 1e-8, 0.05, 7, gs, 0.25)
    pr, mr, vr = adamw_update(p, g * 0.125, m0, v0, 7, 3e-3, 0.05)
    assert close(pd, pr, F32, 1e-6) and close(md, mr, F32, 1e-6) and close(vd, vr, F32, 1e-6)
    part = torch.zeros(2 * 256, device=gpu_device, dtype=torch.float64)
    ops.sumsq(g.to(gpu_device), part, 0)
    ops.sumsq(p.to(gpu_device), part, 1)
    out2 = torch.empty(2, device=gpu_device)
    ops.clip_coef(part, 512, 0.5, 2.0, out2)
    nrm = 0.5 * math.sqrt((g.double() ** 2).sum() + (p.double() ** 2).sum())
    assert abs(out2[0].item() - nrm) < 1e-4 * nrm and abs(out2[1].item() - min(1.0, 2.0 / (nrm + 1e-6))) < 1e-6
    src = rnd(5, 64, seed=5)
    idx = torch.tensor([3, 1, 3, 0, 3], dtype=torch.int32)
    dst = torch.zeros(6, 64, device=gpu_device)
    ops.scatter_add_rows(src.to(gpu_device), 64, idx.to(gpu_device), dst, 5, 64)
    assert close(dst, torch.zeros(6, 64).index_add_(0, idx.long(), src), F32, 1e-6)
    cond = rnd(4, 64, seed=6).requires_grad_(True)
    ds = rnd(4, 64, seed=7)
    F.silu(cond).backward(ds)
    dc = torch.empty(4, 64, device=gpu_device)
    ops.silu_bwd(cond.detach().to(gpu_device), ds.to(gpu_device), dc)
    assert close(dc, cond.grad, F32, 1e-5)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('n,c', [(1000, 200), (64, 64), (37, 130), (4097, 96)])
def test_transpose_tiles_and_rowsum(gpu_device, dtype, n, c):
    B = 2
    x = rnd(B, n, c + 8, seed=1)
    xd = dev(x, dtype, gpu_device)
    npad = (n + 7) // 8 * 8
    out = torch.zeros(B, c, npad, device=gpu_device, dtype=dtype)
    ops.transpose(xd, out, B, n, c, c + 8, in_off=8, ld_out=npad)
    assert torch.equal(out[:, :, :n].float().cpu(), xd.float().cpu()[:, :, 8:].transpose(1, 2))
    assert npad == n or out[:, :, n:].abs().max() == 0
    rs = torch.ones(c, device=gpu_device)
    ops.rowsum(out[0], npad, rs, c, n, accumulate=True)
    assert close(rs, 1 + xd.float().cpu()[0, :, 8:].sum(0), F32, 2e-4)


@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('M,N,K', [(300, 200, 96), (4096, 512, 128), (512, 256, 2048)])   # scalar epilogue / specialised 256x256 / split-K
def test_gemm_fused_gelu_forward_keeps_preactivation_and_gelu_grad_epilogue(gpu_device, dtype, M, N, K):
    """fc1 forward of training: C = gelu(A W^T + b) with the pre-activation stored next to it; fc2 dgrad: C = (A W^T) * gelu'(aux)."""
    from controlvar_amd._lib import ACT_GELU_GRAD, ACT_GELU_TANH
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3)
    ar, wr = a.to(dtype).float(), w.to(dtype).float()
    pre_ref = ar @ wr.t() + b
    ad, wd = dev(a, dtype, gpu_device), dev(w, dtype, gpu_device)
    out = torch.empty(M, N, device=gpu_device, dtype=dtype)
    pre = torch.empty(M, N, device=gpu_device, dtype=dtype)
    ops.gemm(ad, wd, out, M=M, N=N, K=K, bias=b.to(gpu_device), act=ACT_GELU_TANH, pre_act=pre)
    assert close(pre, pre_ref, dtype)
    assert close(out, F.gelu(pre_ref, approximate='tanh'), dtype)
    # gelu' epilogue against autograd of the same function, with the stored (rounded) pre-activation as the operand
    aux = pre.float().cpu().requires_grad_(True)
    F.gelu(aux, approximate='tanh').sum().backward()
    dref = (ar @ wr.t()) * aux.grad
    dh = torch.empty(M, N, device=gpu_device, dtype=dtype)
    ops.gemm(ad, wd, dh, M=M, N=N, K=K, act=ACT_GELU_GRAD, aux=pre)
    assert close(dh, dref, dtype, f32_tol=2e-4)
    with pytest.raises(Exception):
        ops.gemm(ad, wd, dh, M=M, N=N, K=K, act=ACT_GELU_GRAD)                 # aux missing -> CVAR_EINVAL
    with pytest.raises(Exception):
        ops.gemm(ad, wd, dh, M=M, N=N, K=K, act=ACT_GELU_GRAD, aux=pre, pre_act=pre)       # pre_act has no meaning with the gelu' epilogue
    # proj / fc2 forward of training: x1 = x + (gate * keep) * f with f = A W^T + b stored alongside (operand dtype)
    rows = 100
    ng = (M + rows - 1) // rows
    gate, keep, x = rnd(ng, N, seed=5), torch.rand(ng, generator=torch.Generator().manual_seed(6)) + 0.5, rnd(M, N, seed=7)
    x1 = torch.empty(M, N, device=gpu_device, dtype=F32)
    f = torch.empty(M, N, device=gpu_device, dtype=dtype)
    ops.gemm(ad, wd, x1, M=M, N=N, K=K, bias=b.to(gpu_device), gate=gate.to(gpu_device), ldg=N, gate_rows=rows, gate_scale=keep.to(gpu_device),
             residual=x.to(gpu_device), pre_act=f)
    assert close(f, pre_ref, dtype)
    g_rows = (gate * keep[:, None]).repeat_interleave(rows, 0)[:M]
    assert close(x1, x + g_rows * pre_ref, F32 if dtype == F32 else BF16)
    with pytest.raises(Exception):
        ops.gemm(ad, wd, x1, M=M, N=N, K=K, gate_scale=keep.to(gpu_device))   # gate_scale without a gate -> CVAR_EINVAL


@pytest.mark.parametrize('T,Nn,Kk,lda,ldb', [(96, 128, 256, 128, 256),            # three whole K steps, one tile
                                            (1360, 384, 512, 384, 512),           # several tiles, token split through the workspace
                                            (1000, 256, 256, 640, 768),           # T % 32 != 0 (zero-filled tail step), operands are column windows
                                            (50, 128, 256, 128, 256),             # fewer tokens than two K steps
                                            (777, 256, 384, 256, 384),            # Kk ends in the middle of a 256-wide column tile (ldb == Kk: the idle half reads the next token's row)
                                            (1360, 384, 128, 392, 144),           # a single half-filled column tile, NaN columns right behind the window
                                            (1360, 512, 768, 520, 768),           # two row tiles of the 256-row (eight-wave) instance, token split, lda > Nn
                                            (43520, 1536, 1536, 1536, 1536),      # the d24 proj weight gradient at B = 32
                                            (10880, 1920, 1920, 1920, 1920)])     # d30 (C = 1920 = 7.5 column tiles) at B = 8
def test_gemm_tn_weight_gradient(gpu_device, T, Nn, Kk, lda, ldb):
    """cvar_gemm_tn: dW[n, k] = sum_t dY[t, n] X[t, k] read token-major (LDS transpose-read, no transposed copies) against torch on the
    bf16-rounded operands.  The operands sit inside NaN-filled buffers and, where lda > Nn, inside wider NaN rows: every element outside
    the (T x Nn) / (T x Kk) windows that reached an accumulator would poison the result."""
    g = torch.Generator().manual_seed(T + Nn)
    A = torch.randn(T, Nn, generator=g).to(torch.bfloat16)
    B = torch.randn(T, Kk, generator=g).to(torch.bfloat16)
    pad = 4096

    def fenced(t, ld):
        rows = torch.full((T, ld), float('nan'), dtype=torch.bfloat16)
        rows[:, :t.shape[1]] = t
        buf = torch.full((T * ld + 2 * pad,), float('nan'), dtype=torch.bfloat16, device=gpu_device)
        v = buf[pad:pad + T * ld].view(T, ld)
        v.copy_(rows)
        return v
    Ad, Bd = fenced(A, lda), fenced(B, ldb)
    out = torch.full((Nn + 2, Kk + 8), float('nan'), device=gpu_device)
    ops.gemm_tn(Ad, Bd, out, T=T, Nn=Nn, Kk=Kk, lda=lda, ldb=ldb, ldc=Kk + 8)
    got = out[:Nn, :Kk].cpu()
    assert torch.isnan(out[Nn:]).all() and torch.isnan(out[:, Kk:]).all()            # nothing written outside the result window
    if T <= 2000:
        ref = A.double().t() @ B.double()
        assert torch.isfinite(got).all()
        assert ((got.double() - ref).abs() <= 1e-4 * (ref.abs() + math.sqrt(T))).all(), (got.double() - ref).abs().max().item()
    else:                                                                            # full size: against the round-1 path (two transposes + cvar_gemm)
        TA = torch.zeros(Nn, T, device=gpu_device, dtype=torch.bfloat16); TB = torch.zeros(Kk, T, device=gpu_device, dtype=torch.bfloat16)
        ops.transpose(Ad, TA, 1, T, Nn, lda, ld_out=T); ops.transpose(Bd, TB, 1, T, Kk, ldb, ld_out=T)
        old = torch.empty(Nn, Kk, device=gpu_device)
        ops.gemm(TA, TB, old, M=Nn, N=Kk, K=T)
        assert torch.isfinite(got).all()
        assert ((got - old.cpu()).abs() <= 2e-5 * (old.cpu().abs() + math.sqrt(T))).all()
    # run-to-run bit-reproducible (the token split is summed in a fixed order); round 3: the bias gradient (column sums of A) from the same pass -
    # it must not change dW, must equal the exact column sums of the bf16 operand, and must stay inside its Nn floats
    out2 = torch.empty_like(out)
    csbuf = torch.full((Nn + 64,), float('nan'), device=gpu_device)
    ops.gemm_tn(Ad, Bd, out2, T=T, Nn=Nn, Kk=Kk, lda=lda, ldb=ldb, ldc=Kk + 8, colsum=csbuf, colsum_off=32)
    assert torch.equal(out2[:Nn, :Kk].cpu(), got)
    assert torch.isnan(csbuf[:32]).all() and torch.isnan(csbuf[32 + Nn:]).all()
    cs_ref = A.double().sum(0)
    cs = csbuf[32:32 + Nn].double().cpu()
    assert torch.isfinite(cs).all() and ((cs - cs_ref).abs() <= 1e-5 * (cs_ref.abs() + math.sqrt(T))).all(), (cs - cs_ref).abs().max().item()
    csbuf2 = torch.full_like(csbuf, float('nan'))
    ops.gemm_tn(Ad, Bd, out2, T=T, Nn=Nn, Kk=Kk, lda=lda, ldb=ldb, ldc=Kk + 8, colsum=csbuf2, colsum_off=32)
    assert torch.equal(csbuf2[32:32 + Nn], csbuf[32:32 + Nn])
    from controlvar_amd._lib import CvarError
    with pytest.raises(CvarError):
        ops.gemm_tn(Ad, Bd, out, T=T, Nn=Nn - 64, Kk=Kk, lda=lda, ldb=ldb, ldc=Kk + 8)      # Nn must be a multiple of the 128-row tile


@pytest.mark.parametrize('B,n,rows,skip,C', [(3, 50, 52, 2, 128), (4, 1358, 1360, 2, 192), (2, 7, 7, 0, 64), (32, 1358, 1360, 2, 1536)])
def test_wordembed_grad_against_torch(gpu_device, B, n, rows, skip, C):
    """cvar_wordembed_grad (ABI 14): dW = dX^T tok and db = column sums of dX over the word-embedded rows of every sample (the first `skip` rows
    of a sample are not), straight from the token-major fp32 tensors, against float64; the skipped rows are NaN so that a wrong row map shows."""
    g = torch.Generator().manual_seed(B * 1000 + n)
    dx = torch.randn(B, rows, C, generator=g)
    tok = torch.randn(B * n, 32, generator=g)
    dxd = dx.clone()
    dxd[:, :skip] = float('nan')
    dxd = dxd.to(gpu_device)
    out = torch.full((C * 32 + C + 16,), float('nan'), device=gpu_device)
    ops.wordembed_grad(dxd.view(B * rows, C), C, rows, skip, tok.to(gpu_device), n, B, C, 32, out, 8, 8 + C * 32)
    used = dx[:, skip:skip + n].reshape(B * n, C).double()
    dW = used.t() @ tok.double()
    db = used.sum(0)
    got_w, got_b = out[8:8 + C * 32].view(C, 32).double().cpu(), out[8 + C * 32:8 + C * 32 + C].double().cpu()
    assert torch.isnan(out[:8]).all() and torch.isnan(out[8 + C * 32 + C:]).all()
    tol = 2e-6 * math.sqrt(B * n)
    assert (got_w - dW).abs().max() < tol * max(1.0, dW.abs().max().item()) and (got_b - db).abs().max() < tol * max(1.0, db.abs().max().item())
    out2 = torch.empty_like(out)
    ops.wordembed_grad(dxd.view(B * rows, C), C, rows, skip, tok.to(gpu_device), n, B, C, 32, out2, 8, 8 + C * 32)
    assert torch.equal(out2[8:8 + C * 33], out[8:8 + C * 33])                    # fixed summation order
