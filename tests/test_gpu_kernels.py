"""Per-kernel numerics of libcvar_hip.so on a real MI355X, each against a plain torch fp32
computation of the same op (CPU).  All calls go through the C ABI (controlvar_amd.ops)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from controlvar_amd import ops  # noqa: E402
from controlvar_amd._lib import ACT_GELU_TANH  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_dev(t, dtype, dev):
    return t.to(dtype).to(dev).contiguous()


def close(got, ref, dtype, f32_tol=1e-4, bf16_rel=1e-2):
    """fp32: absolute tolerance; bf16 outputs: 1e-2 * (|ref| + 1) (bf16 has 8 mantissa bits)"""
    got, ref = got.float().cpu(), ref.float()
    if dtype == torch.float32:
        return bool(((got - ref).abs() <= f32_tol * (ref.abs() + 1)).all())
    return bool(((got - ref).abs() <= bf16_rel * (ref.abs() + 1)).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(200, 130, 72), (256, 256, 256), (64, 384, 128), (3, 5, 8), (1000, 96, 1536)])
def test_gemm_plain(gpu_device, dtype, M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    Ad, Wd = to_dev(A, dtype, gpu_device), to_dev(W, dtype, gpu_device)
    out = torch.empty(M, N, device=gpu_device, dtype=torch.float32)
    ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=b.to(gpu_device))
    ref = Ad.float().cpu() @ Wd.float().cpu().t() + b
    err = (out.cpu() - ref).abs().max().item()
    assert err < (1e-4 if dtype == torch.float32 else 2e-3) * math.sqrt(K), err


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_epilogues(gpu_device, dtype):
    M, N, K, l = 192, 160, 64, 48           # 4 sequences of 48 rows
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    gate = rnd(M // l, 3 * N, seed=4)        # gate lives at column offset N of a wider table
    x = rnd(M, N, seed=5)
    Ad, Wd = to_dev(A, dtype, gpu_device), to_dev(W, dtype, gpu_device)
    acc = Ad.float().cpu() @ Wd.float().cpu().t() + b
    # bias + gelu -> compute dtype
    out = torch.empty(M, N, device=gpu_device, dtype=dtype)
    ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=b.to(gpu_device), act=ACT_GELU_TANH)
    ref = F.gelu(acc, approximate='tanh')
    assert close(out, ref, dtype)
    # gated residual in place (fp32 residual stream)
    xd = x.to(gpu_device).clone()
    ops.gemm(Ad, Wd, xd, M=M, N=N, K=K, bias=b.to(gpu_device), gate=gate.to(gpu_device), gate_off=N, ldg=3 * N, gate_rows=l, residual=xd)
    g = gate[:, N:2 * N].repeat_interleave(l, dim=0)
    ref = x + acc * g
    assert close(xd, ref, dtype, 2e-4)
    # row remap into an arena [R][Lmax][N]
    R, Lmax, off = M // l, 100, 7
    arena = torch.zeros(R, Lmax, N, device=gpu_device, dtype=dtype)
    ops.gemm(Ad, Wd, arena, M=M, N=N, K=K, bias=b.to(gpu_device), remap=(l, Lmax, off))
    got = arena[:, off:off + l].reshape(M, N).float().cpu()
    assert close(got, acc, dtype)
    assert arena[:, :off].abs().max() == 0 and arena[:, off + l:].abs().max() == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_batched_strided(gpu_device, dtype):
    B, n, c = 3, 96, 64
    qkv = rnd(B, n, 3 * c, seed=7)
    qd = to_dev(qkv, dtype, gpu_device)
    s = torch.empty(B, n, n, device=gpu_device, dtype=torch.float32)
    ops.gemm(qd, qd, s, M=n, N=n, K=c, lda=3 * c, ldw=3 * c, w_off=c, alpha=0.125, batch=B, strideA=n * 3 * c, strideW=n * 3 * c, strideC=n * n)
    q, k = qd.float().cpu()[..., :c], qd.float().cpu()[..., c:2 * c]
    ref = torch.bmm(q, k.transpose(1, 2)) * 0.125
    assert close(s, ref, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('mode', ['s1', 'up', 's2'])
@pytest.mark.parametrize('cin,cout', [(32, 48), (160, 160), (8, 24)])
def test_conv3x3(gpu_device, dtype, mode, cin, cout):
    B, H, W = 2, 12, 10
    x = rnd(B, cin, H, W, seed=11)
    w = rnd(cout, cin, 3, 3, seed=12, scale=1.0 / math.sqrt(9 * cin))
    b = rnd(cout, seed=13)
    res = None
    xr = x.to(dtype).float()
    wr = w.to(dtype).float()
    if mode == 's1':
        ref = F.conv2d(xr, wr, b, padding=1); Ho, Wo = H, W
        res = rnd(B, cout, H, W, seed=14)
        ref = ref + res.to(dtype).float()
    elif mode == 'up':
        ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, b, padding=1); Ho, Wo = 2 * H, 2 * W
    else:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, b, stride=2, padding=0); Ho, Wo = H // 2, W // 2
    xd = to_dev(x.permute(0, 2, 3, 1).reshape(B * H * W, cin), dtype, gpu_device)
    wd = to_dev(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin), dtype, gpu_device)
    resd = to_dev(res.permute(0, 2, 3, 1).reshape(B * Ho * Wo, cout), dtype, gpu_device) if res is not None else None
    out = torch.empty(B * Ho * Wo, cout, device=gpu_device, dtype=torch.float32)
    ops.gemm(xd, wd, out, M=B * Ho * Wo, N=cout, K=9 * cin, bias=b.to(gpu_device), residual=resd,
             conv=dict(Hin=H, Win=W, Cin=cin, Hout=Ho, Wout=Wo, stride=2 if mode == 's2' else 1, up=1 if mode == 'up' else 0))
    got = out.cpu().reshape(B, Ho, Wo, cout).permute(0, 3, 1, 2)
    assert close(got, ref, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cin,cout,up', [(32, 160, 0), (160, 320, 1), (160, 160, 2)])     # up = 2 stands for the stride-2 downsample
def test_conv3x3_wide_tile(gpu_device, dtype, cin, cout, up):
    """M >= 4096 and N % 160 == 0: exercises the 128x160 tile configuration (VQVAE channel counts)."""
    s2 = up == 2
    up = 0 if s2 else up
    B, H, W = (2, 97, 96) if s2 else (2, 48, 48)             # odd height: the (0,1,0,1) pad row is exercised
    x = rnd(B, cin, H, W, seed=31)
    w = rnd(cout, cin, 3, 3, seed=32, scale=1.0 / math.sqrt(9 * cin))
    b = rnd(cout, seed=33)
    xr, wr = x.to(dtype).float(), w.to(dtype).float()
    Ho, Wo = (2 * H, 2 * W) if up else ((H // 2, W // 2) if s2 else (H, W))
    res = rnd(B, cout, Ho, Wo, seed=34)
    if s2:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, b, stride=2, padding=0)[:, :, :Ho, :Wo] + res.to(dtype).float()
    else:
        src = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
        ref = F.conv2d(src, wr, b, padding=1) + res.to(dtype).float()
    xd = to_dev(x.permute(0, 2, 3, 1).reshape(B * H * W, cin), dtype, gpu_device)
    wd = to_dev(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin), dtype, gpu_device)
    resd = to_dev(res.permute(0, 2, 3, 1).reshape(B * Ho * Wo, cout), dtype, gpu_device)
    out = torch.empty(B * Ho * Wo, cout, device=gpu_device, dtype=dtype)
    ops.gemm(xd, wd, out, M=B * Ho * Wo, N=cout, K=9 * cin, bias=b.to(gpu_device), residual=resd,
             conv=dict(Hin=H, Win=W, Cin=cin, Hout=Ho, Wout=Wo, stride=2 if s2 else 1, up=up))
    got = out.float().cpu().reshape(B, Ho, Wo, cout).permute(0, 3, 1, 2)
    assert close(got, ref, dtype, bf16_rel=2e-2)


@pytest.mark.parametrize('out_dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [128, 768, 1920])
def test_ln_modulate(gpu_device, out_dtype, C):
    R, l = 3, 7
    x = rnd(R * l, C, seed=1, scale=2.0) + 0.5
    ada = rnd(R, 6 * C, seed=2, scale=0.3)
    out = torch.empty(R * l, C, device=gpu_device, dtype=out_dtype)
    ops.ln_modulate(x.to(gpu_device), ada.to(gpu_device), 2 * C, 4 * C, 6 * C, l, out, R * l, C, 1e-6)
    sc = ada[:, 2 * C:3 * C].repeat_interleave(l, 0)
    sh = ada[:, 4 * C:5 * C].repeat_interleave(l, 0)
    ref = F.layer_norm(x, (C,), eps=1e-6) * (1 + sc) + sh
    assert close(out, ref, out_dtype, 2e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('masked', [False, True])
def test_attention(gpu_device, dtype, masked):
    R, H, c = 2, 3, 64
    lvl_end = [2, 10, 28, 60]
    Lmax = 60
    C3 = 3 * H * c
    qkv = rnd(R, Lmax, C3, seed=5)
    qd = to_dev(qkv, dtype, gpu_device)
    qf = qd.float().cpu().view(R, Lmax, 3, H, c)
    scale = 0.03125
    if masked:
        q_off, l = 0, Lmax
    else:
        q_off, l = 28, 32
    out = torch.empty(R * l, H * c, device=gpu_device, dtype=dtype)
    ops.attention(qd, out, R, H, Lmax, q_off, l, scale, lvl_end if masked else None)
    q = qf[:, q_off:q_off + l, 0].permute(0, 2, 1, 3)
    k = qf[:, :q_off + l, 1].permute(0, 2, 1, 3)
    v = qf[:, :q_off + l, 2].permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) * scale
    if masked:
        lvl = torch.cat([torch.full((e - b,), i) for i, (b, e) in enumerate(zip([0] + lvl_end[:-1], lvl_end))])
        s = s + torch.where(lvl.view(-1, 1) >= lvl.view(1, -1), 0., -torch.inf)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(R * l, H * c)
    assert close(out, ref, dtype, 2e-5, 2e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cos_qk_norm(gpu_device, dtype):
    R, H, Lmax, q_off, l = 2, 3, 20, 5, 9
    qkv = rnd(R, Lmax, 3 * H * 64, seed=3)
    sm = torch.tensor([0.2, 1.4, 5.0])
    qd = to_dev(qkv, dtype, gpu_device)
    before = qd.float().cpu().clone()
    ops.cos_qk_norm(qd, R, H, Lmax, q_off, l, sm.to(gpu_device))
    after = qd.float().cpu().view(R, Lmax, 3, H, 64)
    b5 = before.view(R, Lmax, 3, H, 64)
    ref = b5.clone()
    ref[:, q_off:q_off + l, 0] = F.normalize(b5[:, q_off:q_off + l, 0], dim=-1) * sm.clamp_max(math.log(100)).exp().view(1, 1, H, 1)
    ref[:, q_off:q_off + l, 1] = F.normalize(b5[:, q_off:q_off + l, 1], dim=-1)
    assert close(after, ref, dtype, 1e-5, 1e-2)
    assert torch.equal(after[:, :q_off], b5[:, :q_off]) and torch.equal(after[:, :, 2], b5[:, :, 2])


def test_cfg_greedy_and_combine(gpu_device):
    B, l, V = 3, 5, 4096
    logits = rnd(2 * B, l, V, seed=9, scale=3.0)
    t = 4.0 * 3 / 9
    idx = torch.empty(B, l, device=gpu_device, dtype=torch.int32)
    comb = torch.empty(B, l, V, device=gpu_device)
    mg = torch.empty(B, l, device=gpu_device)
    ops.cfg_sample(logits.to(gpu_device), B, 2, l, V, [1 + t, -t], 1, 0.0, 0, 3, 1, idx, comb, mg)
    ref = (1 + t) * logits[:B] - t * logits[B:]
    assert torch.equal(comb.cpu(), ref)                       # same evaluation order -> bit identical
    assert torch.equal(idx.cpu().long(), ref.argmax(-1))
    t2 = ref.topk(2, dim=-1).values
    assert torch.allclose(mg.cpu(), t2[..., 0] - t2[..., 1])
    # 4-branch form
    lg4 = rnd(4 * B, l, V, seed=10)
    c = [1 + 1.0, 0.5 - 1.0, 0.25 - 0.5, -0.25]
    idx4 = torch.empty(4 * B, l, device=gpu_device, dtype=torch.int32)
    ops.cfg_sample(lg4.to(gpu_device), B, 4, l, V, c, 1, 0.0, 0, 0, 4, idx4, comb, None)
    np32 = np.float32
    ref4 = (np32(c[0]) * lg4[:B] + np32(c[1]) * lg4[B:2 * B] + np32(c[2]) * lg4[2 * B:3 * B]) + np32(c[3]) * lg4[3 * B:]
    assert torch.equal(comb.cpu(), ref4)
    assert torch.equal(idx4.cpu().long(), ref4.argmax(-1).repeat(4, 1))


def test_cfg_sample_topk_topp(gpu_device):
    """kept-set size equals the reference filter's; draws stay inside the kept set and follow its distribution."""
    from oracle.var_ref import topk_topp_mask_
    B, l, V = 2, 4, 4096
    logits = rnd(2 * B, l, V, seed=21, scale=2.5)
    t = 1.7
    ref = (1 + t) * logits[:B] - t * logits[B:]
    for (k, p) in [(900, 0.96), (0, 0.5), (50, 0.0)]:
        masked = topk_topp_mask_(ref.clone(), k, p)
        kept_ref = torch.isfinite(masked)
        kept = torch.empty(B, l, device=gpu_device, dtype=torch.int32)
        idx = torch.empty(B, l, device=gpu_device, dtype=torch.int32)
        counts = torch.zeros(B, l, V)
        n_rounds = 300
        for s in range(n_rounds):
            ops.cfg_sample(logits.to(gpu_device), B, 2, l, V, [1 + t, -t], k, p, 1234 + s, 2, 1, idx, None, None, kept)
            i = idx.cpu().long()
            assert kept_ref.gather(-1, i.unsqueeze(-1)).all(), 'draw outside the reference kept set'
            counts.scatter_add_(-1, i.unsqueeze(-1), torch.ones(B, l, 1))
        # kept-set size: equal to the reference filter's, except where the nucleus threshold falls within fp32 rounding of a cumulative
        # probability (torch's fp32 cumsum order is device-dependent; the kernel accumulates in double) - then one token either way
        dk = kept.cpu().long() - kept_ref.sum(-1)
        assert dk.abs().max() <= 1
        if dk.abs().max() > 0:
            cs = ref.double().sort(-1, descending=False)[0].softmax(-1).cumsum(-1)            # (B, l, V) ascending
            n_rm = V - kept_ref.sum(-1)                                                        # tokens the reference removed
            for b_, t_ in zip(*torch.nonzero(dk, as_tuple=True)):
                near = cs[b_, t_, max(int(n_rm[b_, t_]) - 1, 0):int(n_rm[b_, t_]) + 1]
                assert ((near - (1 - p)).abs() < 1e-5).any(), f'kept-set differs away from the threshold: {near.tolist()} vs {1 - p}'
        probs = masked.softmax(-1)
        top = probs.argmax(-1, keepdim=True)
        p_top = probs.gather(-1, top).squeeze(-1)
        f_top = counts.gather(-1, top).squeeze(-1) / n_rounds
        assert (f_top - p_top).abs().max() < 5 * (p_top * (1 - p_top) / n_rounds).sqrt().max() + 0.02


def test_word_embed_first_tokens(gpu_device):
    nb, l, Cv, C = 3, 8, 32, 128
    tok, W, b = rnd(nb, l, Cv, seed=1), rnd(C, Cv, seed=2), rnd(C, seed=3)
    lvl = rnd(40, C, seed=4)
    x = torch.zeros(2 * nb, l, C, device=gpu_device)
    ops.word_embed(tok.to(gpu_device), W.to(gpu_device), b.to(gpu_device), lvl.to(gpu_device), x, nb, 2, l, Cv, C, l, 0, lvl_off=10)
    ref = F.linear(tok, W, b) + lvl[10:10 + l]
    assert (x.cpu() - ref.repeat(2, 1, 1)).abs().max() < 1e-5
    R = 4
    ce, cd = rnd(1001, C, seed=5), rnd(5, C, seed=6)
    ps = rnd(2, C, seed=7)
    labels = torch.tensor([3, 1000, 999, 0], dtype=torch.int32)
    types = torch.tensor([0, 4, 2, 3], dtype=torch.int32)
    x0 = torch.empty(R, 2, C, device=gpu_device)
    cond = torch.empty(R, C, device=gpu_device)
    ops.first_tokens(ce.to(gpu_device), cd.to(gpu_device), labels.to(gpu_device), types.to(gpu_device), ps.to(gpu_device), lvl.to(gpu_device), x0, cond, R, 2, C, 2)
    ref0 = torch.stack([cd[types.long()], ce[labels.long()]], 1) + ps + lvl[:2]
    assert (x0.cpu() - ref0).abs().max() < 1e-6 and torch.equal(cond.cpu(), ce[labels.long()])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,HW', [(160, 1024), (640, 256), (32, 4096)])
def test_groupnorm_silu(gpu_device, dtype, C, HW):
    B = 2
    x = rnd(B, HW, C, seed=1, scale=1.5) + 0.7
    w, b = rnd(C, seed=2, scale=0.1) + 1, rnd(C, seed=3, scale=0.1)
    xd = to_dev(x, dtype, gpu_device)
    out = torch.empty_like(xd)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=gpu_device, dtype=torch.uint8)
    ops.groupnorm_silu(xd, w.to(gpu_device), b.to(gpu_device), out, B, HW, C, 32, 1e-6, True, ws)
    ref = F.silu(F.group_norm(xd.float().cpu().permute(0, 2, 1), 32, w, b, eps=1e-6)).permute(0, 2, 1)
    assert close(out, ref, dtype, 2e-5, 2e-2)


def test_softmax_transpose_layout(gpu_device):
    s = rnd(12, 256, seed=1, scale=3)
    p = torch.empty(12, 256, device=gpu_device)
    ops.softmax_rows(s.to(gpu_device), p, 12, 256)
    assert (p.cpu() - s.softmax(-1)).abs().max() < 1e-6
    x = rnd(2, 50, 3 * 40, seed=2)
    out = torch.empty(2, 40, 50, device=gpu_device)
    ops.transpose(x.to(gpu_device), out, 2, 50, 40, 120, in_off=80)
    assert torch.equal(out.cpu(), x[:, :, 80:].transpose(1, 2))
    img = rnd(2, 3, 64, seed=3)
    o = torch.empty(2 * 64, 8, device=gpu_device, dtype=torch.bfloat16)
    ops.nchw_to_nhwc(img.to(gpu_device), o, 2, 3, 64, 8)
    oc = o.float().cpu().view(2, 64, 8)
    assert torch.equal(oc[..., :3], img.to(torch.bfloat16).float().permute(0, 2, 1)) and oc[..., 3:].abs().max() == 0
    back = torch.empty(2, 3, 64, device=gpu_device)
    ops.nhwc_to_nchw(o, 8, back, 2, 3, 64, -0.5, 0.5, 0.5, 0.5)
    assert torch.allclose(back.cpu(), img.to(torch.bfloat16).float().clamp(-0.5, 0.5) * 0.5 + 0.5)


@pytest.mark.parametrize('q_off,l,masked', [(0, 2, False), (10, 18, False), (310, 200, False), (848, 512, False), (0, 1360, True)])
def test_attention_mfma_flash_bf16(gpu_device, q_off, l, masked):
    """bf16 MFMA flash kernel vs (a) the exact row-wise kernel of the same library and (b) torch fp32 math."""
    from controlvar_amd.spec import Pyramid
    py = Pyramid()
    R, H, c, Lmax = 2, 2, 64, 1360
    C3 = 3 * H * c
    qkv = rnd(R, Lmax, C3, seed=5, scale=1.5)
    qd = to_dev(qkv, torch.bfloat16, gpu_device)
    scale = 0.03125 * 4
    lvl_end = list(py.end) if masked else None
    out = torch.empty(R * l, H * c, device=gpu_device, dtype=torch.bfloat16)
    out_rw = torch.empty_like(out)
    ops.attention(qd, out, R, H, Lmax, q_off, l, scale, lvl_end)
    ops.attention(qd, out_rw, R, H, Lmax, q_off, l, scale, lvl_end, rowwise=True)
    qf = qd.float().cpu().view(R, Lmax, 3, H, c)
    q = qf[:, q_off:q_off + l, 0].permute(0, 2, 1, 3)
    k = qf[:, :q_off + l, 1].permute(0, 2, 1, 3)
    v = qf[:, :q_off + l, 2].permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2) * scale
    if masked:
        lvl = torch.from_numpy(py.level_of_token())
        s = s + torch.where(lvl.view(-1, 1) >= lvl.view(1, -1), 0., -torch.inf)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(R * l, H * c)
    assert close(out_rw, ref, torch.bfloat16, bf16_rel=2e-2)
    assert close(out, ref, torch.bfloat16, bf16_rel=2e-2)
    assert (out.float() - out_rw.float()).abs().max() < 2e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(4, 1536, 1536), (256, 1536, 6144), (1000, 768, 1024), (36, 4608, 1536)])
def test_gemm_split_k(gpu_device, dtype, M, N, K):
    """small-M GEMMs take the split-K route once a workspace is registered: partial tiles + fixed-order reduce + epilogue"""
    ops.ensure_splitk_workspace(gpu_device)
    l = 4 if M % 4 == 0 else 1
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / math.sqrt(K)), rnd(N, seed=3)
    gate = rnd(max(1, M // l), N, seed=4)
    x = rnd(M, N, seed=5)
    Ad, Wd = to_dev(A, dtype, gpu_device), to_dev(W, dtype, gpu_device)
    acc = Ad.float().cpu() @ Wd.float().cpu().t() + b
    out = torch.empty(M, N, device=gpu_device, dtype=dtype)
    ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=b.to(gpu_device), act=ACT_GELU_TANH)
    assert close(out, F.gelu(acc, approximate='tanh'), dtype, 2e-4)
    xd = x.to(gpu_device).clone()
    ops.gemm(Ad, Wd, xd, M=M, N=N, K=K, bias=b.to(gpu_device), gate=gate.to(gpu_device), ldg=N, gate_rows=l, residual=xd)
    assert close(xd, x + acc * gate.repeat_interleave(l, 0)[:M], dtype, 3e-4)
    R, Lmax, off = M // l, 3 * l + 5, 2
    arena = torch.zeros(R, Lmax, N, device=gpu_device, dtype=dtype)
    ops.gemm(Ad, Wd, arena, M=M, N=N, K=K, bias=b.to(gpu_device), remap=(l, Lmax, off))
    assert close(arena[:, off:off + l].reshape(M, N), acc, dtype, 2e-4) and arena[:, :off].abs().max() == 0


def test_conv_fast_never_reads_past_the_weights(gpu_device):
    """K = 9*Cin is not a multiple of the 64-element K tile: the tail of the LAST weight row lies past the tensor.  The
    activations are zero there, but 0 x NaN is NaN - the weights are placed directly in front of NaNs to prove the kernel's
    buffer range stops at the last weight (regression: decoder output depended on what the allocator had left behind)."""
    dtype = torch.bfloat16
    B, H, W, cin, cout = 2, 48, 48, 160, 160
    x = rnd(B, cin, H, W, seed=41)
    w = rnd(cout, cin, 3, 3, seed=42, scale=1.0 / math.sqrt(9 * cin))
    b = rnd(cout, seed=43)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), b, padding=1)
    xd = to_dev(x.permute(0, 2, 3, 1).reshape(B * H * W, cin), dtype, gpu_device)
    n = cout * 9 * cin
    buf = torch.full((n + 4096,), float('nan'), device=gpu_device, dtype=dtype)
    buf[:n] = w.permute(0, 2, 3, 1).reshape(-1).to(dtype).to(gpu_device)
    wd = buf[:n].view(cout, 9 * cin)
    out = torch.empty(B * H * W, cout, device=gpu_device, dtype=torch.float32)
    ops.gemm(xd, wd, out, M=B * H * W, N=cout, K=9 * cin, bias=b.to(gpu_device),
             conv=dict(Hin=H, Win=W, Cin=cin, Hout=H, Wout=W, stride=1, up=0))
    got = out.cpu().reshape(B, H, W, cout).permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert close(got, ref, dtype)


def test_entry_points_reject_bad_arguments_with_a_status(gpu_device):
    """ABI rule: nothing throws or aborts across the boundary - bad arguments come back as a negative cvar_status
    (surfaced by the ctypes layer as CvarError), and the device stays usable afterwards."""
    from controlvar_amd import _lib
    from controlvar_amd._lib import CvarError
    lib = _lib.load()
    x = torch.zeros(64, 64, device=gpu_device, dtype=torch.bfloat16)
    o = torch.zeros(64, 64, device=gpu_device, dtype=torch.float32)
    with pytest.raises(CvarError):
        ops.gemm(x, x, o, M=0, N=64, K=64)                                                   # empty problem
    with pytest.raises(TypeError):
        ops.gemm(x, x.float(), o, M=64, N=64, K=64)                                           # operand dtypes differ (host check)
    with pytest.raises(CvarError):
        ops.gemm(x, x, o, M=64, N=64, K=64, gate=o, ldg=64, gate_rows=0)                      # gate without rows
    with pytest.raises(CvarError):
        ops.attention(x, x, 1, 1, 64, 60, 8, 0.1, None)                                       # q_off + l > Lmax
    with pytest.raises(CvarError):
        ops.ln_modulate(o, o, 0, 0, 64, 0, x, 64, 64, 1e-6)                                   # rows_per == 0
    assert lib.cvar_gemm(None, None) < 0 and lib.cvar_status_str(-1) and lib.cvar_status_str(0)
    # the device is still fine
    ops.gemm(x, x, o, M=64, N=64, K=64)
    torch.cuda.synchronize()
    assert torch.equal(o, torch.zeros_like(o))


def test_gemm_randomised_sweep_inside_nan_arenas(gpu_device):
    """60 random GEMM / conv problems (awkward shapes, every epilogue combination, both dtypes) with all operands embedded in
    NaN-filled buffers: a single byte read outside a tensor, or a wrong tile edge, turns the output non-finite or off
    (tools/fuzz_gemm.py is the same sweep at any size; 1100 cases were clean when this was written)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_gemm.py'), '60', '3'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '60/60 cases ok' in r.stdout


def test_attention_randomised_sweep_with_poisoned_keys(gpu_device):
    """80 random attention problems (inference spans and training level masks) on arenas whose invisible keys and surroundings
    are NaN: the MFMA kernel against the exact row-wise kernel of the same library (tools/fuzz_attn.py, 600 cases clean)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_attn.py'), '80', '5'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_attention_backward_randomised_sweep_inside_nan_arenas(gpu_device):
    """40 random training-attention problems (ragged block-causal level structures): the MFMA backward against the exact
    row-wise backward of the same library, every tensor surrounded by NaN pads that must stay intact (tools/fuzz_attn_bwd.py)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_attn_bwd.py'), '40', '3'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '40/40 cases ok' in r.stdout


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_groupnorm_large_mean_small_spread(gpu_device, dtype):
    """|mean| >> std inside a group: a raw sum / sum-of-squares pass loses every digit of the variance to cancellation (round-1 limit:
    2e-3 .. 7e-2 on such groups).  The statistics are accumulated around a per-channel pivot and reassembled in double, so the
    normalised output must match a float64 GroupNorm of the SAME stored values."""
    B, HW, C, G = 2, 96, 64, 32
    g = torch.Generator().manual_seed(12)
    x = (300.0 + 50.0 * torch.randn(B, 1, C, generator=g)) + 0.05 * torch.randn(B, HW, C, generator=g)          # per-channel offsets of O(300), spread 0.05
    xd = x.reshape(B * HW, C).to(dtype).to(gpu_device).contiguous()
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    out = torch.empty_like(xd)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=gpu_device, dtype=torch.uint8)
    ops.groupnorm_silu(xd, w.to(gpu_device), b.to(gpu_device), out, B, HW, C, G, 1e-6, False, ws)
    xs = xd.double().cpu().view(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xs, G, w.double(), b.double(), 1e-6).permute(0, 2, 1).reshape(B * HW, C)
    err = (out.double().cpu() - ref).abs().max().item()
    assert err < (2e-3 if dtype == torch.float32 else 3e-2), err          # fp32: output rounding of O(1) values scaled by rstd ~ 1e-2; bf16: output storage


# ------------------------------------------------------------------------------------------------ K/V arena form (ABI 11)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('R,l,C,K', [(2, 9, 128, 64),        # small-M: split-K path
                                     (4, 100, 1920, 256),    # d30 width: the q | k boundary (1920) crosses a 256-wide tile
                                     (8, 256, 768, 768),     # 256x256 tiles, boundary on a tile edge
                                     (3, 700, 256, 128)])    # ragged M
def test_gemm_column_split_equals_the_unsplit_remap(gpu_device, dtype, R, l, C, K):
    """cvar_gemm_desc.C_split: the q columns of the qkv GEMM go to their own (M, C) buffer, k | v into a [R][Lmax][2C] arena -
    every element BIT-identical to the packed [R][Lmax][3C] result of the same call without the split."""
    M, N, Lmax, off = R * l, 3 * C, l + 37, 11
    A, W, b = to_dev(rnd(M, K, seed=1), dtype, gpu_device), to_dev(rnd(N, K, seed=2), dtype, gpu_device), rnd(N, seed=3).to(gpu_device)
    packed = torch.zeros(R, Lmax, N, device=gpu_device, dtype=dtype)
    ops.gemm(A, W, packed, M=M, N=N, K=K, bias=b, remap=(l, Lmax, off))
    kv = torch.full((R, Lmax, 2 * C), float('nan'), device=gpu_device, dtype=dtype)
    q = torch.full((M, C), float('nan'), device=gpu_device, dtype=dtype)
    ops.gemm(A, W, kv, M=M, N=N, K=K, bias=b, ldc=2 * C, remap=(l, Lmax, off), split=(q, C, C))
    assert torch.equal(q.view(R, l, C), packed[:, off:off + l, :C])
    assert torch.equal(kv[:, off:off + l], packed[:, off:off + l, C:])
    assert torch.isnan(kv[:, :off]).all() and torch.isnan(kv[:, off + l:]).all()          # rows of other scales untouched
    ref = A.float().cpu() @ W.float().cpu().t() + b.cpu()
    assert close(q, ref[:, :C], dtype, 2e-4)
    from controlvar_amd._lib import CvarError
    with pytest.raises(CvarError):
        ops.gemm(A, W, kv, M=M, N=N, K=K, bias=b, ldc=2 * C, split=(q, C, C))                # the split rides on the row remap
    with pytest.raises(CvarError):
        ops.gemm(A, W, kv, M=M, N=N, K=K, ldc=2 * C, remap=(l, Lmax, off), split=(q, C + 4, C + 8))   # not a multiple of 8


@pytest.mark.parametrize('dtype,rowwise', [(torch.float32, True), (torch.bfloat16, True), (torch.bfloat16, False)])
@pytest.mark.parametrize('H,Lmax,q_off,l,levels', [(3, 60, 0, 60, (2, 10, 28, 60)), (2, 700, 188, 512, None), (12, 1360, 848, 512, None), (1, 40, 39, 1, None)])
def test_attention_kv_arena_form_is_bit_identical_to_the_packed_form(gpu_device, dtype, rowwise, H, Lmax, q_off, l, levels):
    """cvar_attention(qkv = K/V arena [R][Lmax][2C], q = [R][l][C]) == the packed-arena call on the same numbers; the separate
    query buffer and every key row the queries must not see are surrounded / filled with NaNs."""
    R, C = 2, H * 64
    qkv = to_dev(rnd(R, Lmax, 3 * C, seed=5), dtype, gpu_device)
    want = torch.empty(R * l, C, device=gpu_device, dtype=dtype)
    ops.attention(qkv, want, R, H, Lmax, q_off, l, 0.125, levels, rowwise=rowwise)
    kv = qkv[:, :, C:].contiguous()
    kv[:, q_off + l:] = float('nan')
    buf = torch.full((R * l * C + 4096,), float('nan'), device=gpu_device, dtype=dtype)
    q = buf[2048:2048 + R * l * C].view(R, l, C)
    q.copy_(qkv[:, q_off:q_off + l, :C])
    got = torch.empty(R * l, C, device=gpu_device, dtype=dtype)
    lse_a = torch.empty(R, H, l, device=gpu_device, dtype=torch.float32)
    lse_b = torch.empty_like(lse_a)
    ops.attention(kv, got, R, H, Lmax, q_off, l, 0.125, levels, rowwise=rowwise, q=q, lse=lse_b)
    ops.attention(qkv, want, R, H, Lmax, q_off, l, 0.125, levels, rowwise=rowwise, lse=lse_a)
    assert torch.isfinite(got.float()).all() and torch.equal(got, want) and torch.equal(lse_a, lse_b)
    # in place: the output may overwrite the query buffer (each (sequence, head, query block) is read and written by one workgroup)
    ops.attention(kv, q.view(R * l, C), R, H, Lmax, q_off, l, 0.125, levels, rowwise=rowwise, q=q)
    assert torch.equal(q.view(R * l, C), want)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cos_qk_norm_kv_arena_form(gpu_device, dtype):
    R, H, Lmax, q_off, l = 2, 3, 20, 5, 9
    C = H * 64
    qkv = to_dev(rnd(R, Lmax, 3 * C, seed=3), dtype, gpu_device)
    sm = torch.tensor([0.2, 1.4, 5.0], device=gpu_device)
    kv = qkv[:, :, C:].contiguous()
    q = qkv[:, q_off:q_off + l, :C].contiguous()
    kv0 = kv.clone()
    ops.cos_qk_norm(qkv, R, H, Lmax, q_off, l, sm)
    ops.cos_qk_norm(kv, R, H, Lmax, q_off, l, sm, q=q)
    assert torch.equal(q, qkv[:, q_off:q_off + l, :C]) and torch.equal(kv, qkv[:, :, C:])
    assert torch.equal(kv[:, :q_off], kv0[:, :q_off]) and torch.equal(kv[:, :, C:], kv0[:, :, C:])      # other rows and V untouched


# ------------------------------------------------------------------------------------------------ LDS-halo 3x3 conv (conv_halo.hip)
@pytest.mark.parametrize('B,H,W,cin,cout,res,up,out_f32', [(3, 32, 48, 32, 160, False, 0, False),      # one channel chunk, non-square, image borders everywhere
                                                           (2, 16, 16, 96, 320, True, 0, False),       # one tile per image, two cout tiles, residual
                                                           (2, 64, 32, 160, 160, True, 0, False),
                                                           (2, 32, 32, 64, 160, False, 1, False),      # behind the nearest x2 upsample (input 16x16)
                                                           (1, 48, 16, 320, 160, True, 1, False),
                                                           (2, 32, 48, 160, 3, False, 0, True),        # conv_out: 3 channels, fp32 output (narrow form)
                                                           (1, 16, 32, 64, 24, False, 0, False),       # narrow form, bf16 output
                                                           (2, 32, 32, 32, 32, False, 1, True)])
def test_conv3x3_halo_kernel_against_torch_and_the_implicit_gemm(gpu_device, B, H, W, cin, cout, res, up, out_f32):
    """conv_halo.hip (forced with tile_cfg 6) against torch's fp32 conv2d of the bf16-rounded operands and against the implicit-GEMM
    tiles (tile_cfg 5).  The images in front of and behind the batch are NaN: a tap that leaves its image must read the zero padding."""
    T = torch.bfloat16
    hin, win = (H // 2, W // 2) if up else (H, W)
    g = torch.Generator().manual_seed(B * 1000 + H + cin)
    xs = torch.randn(B + 2, hin, win, cin, generator=g)
    xs[0] = float('nan'); xs[-1] = float('nan')
    buf = xs.to(T).to(gpu_device)
    x = buf[1:B + 1].reshape(-1, cin)                                   # a view: NaN images on both sides in memory
    w = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(T)
    bias = torch.randn(cout, generator=g)
    r = torch.randn(B * H * W, cout, generator=g).to(T) if res else None
    xin = x.float().cpu().view(B, hin, win, cin).permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode='nearest')
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, cout)
    if res:
        ref = ref + r.float()
    outs = {}
    for cfg in (6, 5):
        out = torch.full((B * H * W + 64, cout), float('nan'), device=gpu_device, dtype=torch.float32 if out_f32 else T)
        ops.GEMM_TILE_CFG = cfg
        try:
            ops.gemm(x, w.reshape(cout, 9 * cin).to(gpu_device), out, M=B * H * W, N=cout, K=9 * cin, bias=bias.to(gpu_device),
                     residual=r.to(gpu_device) if res else None, conv=dict(Hin=hin, Win=win, Cin=cin, Hout=H, Wout=W, up=up))
        finally:
            ops.GEMM_TILE_CFG = 0
        assert torch.isnan(out[B * H * W:].float()).all()               # nothing written behind the tensor
        outs[cfg] = out[:B * H * W].float().cpu()
        assert torch.isfinite(outs[cfg]).all()
        if out_f32:          # bf16 operands, fp32 accumulation and output: only the summation order separates the kernels from torch
            assert ((outs[cfg] - ref).abs() <= 2e-3 * (ref.abs() + 1)).all(), (cfg, (outs[cfg] - ref).abs().max().item())
        else:
            assert close(outs[cfg], ref, T, bf16_rel=1e-2), (cfg, (outs[cfg] - ref).abs().max().item())
    # both kernels round the same fp32 sums (different summation order) to bf16: at most one bf16 step apart
    assert ((outs[6] - outs[5]).abs() <= (1e-4 if out_f32 else 2.0 ** -7) * (ref.abs() + 1)).all()


# ------------------------------------------------------------------------------------------------ round 5: GroupNorm statistics from the conv's epilogue (ABI 18)
@pytest.mark.parametrize('B,H,W,cin,cout,res,up,big_mean', [(2, 32, 32, 160, 160, False, 0, False),
                                                              (3, 16, 48, 320, 320, True, 0, False),      # two 160-channel workgroup columns, residual in the sum
                                                              (1, 64, 32, 160, 160, True, 1, False),      # behind the nearest x2 upsample
                                                              (2, 16, 16, 64, 160, False, 0, True)])      # |mean| >> std: the per-tile pivots carry the precision
def test_groupnorm_from_the_conv_epilogue_partials(gpu_device, B, H, W, cin, cout, res, up, big_mean):
    """cvar_gemm_desc.gn_part (ABI 18): the halo conv writes (sum (y - piv), sum (y - piv)^2, piv) per tile and channel of the output it stores;
    cvar_groupnorm_silu_partials must then give the GroupNorm of that stored tensor - checked against a float64 group_norm of the SAME bf16 values
    (the yardstick of test_groupnorm_large_mean_small_spread) and against the stand-alone statistics pass (<= one bf16 step apart, almost everywhere equal).
    The partials themselves are checked per tile against float64 sums."""
    T = torch.bfloat16
    hin, win = (H // 2, W // 2) if up else (H, W)
    g = torch.Generator().manual_seed(7 * B + H + cin)
    x = torch.randn(B * hin * win, cin, generator=g).to(T).to(gpu_device)
    w = (torch.randn(cout, 9 * cin, generator=g) / (9 * cin) ** 0.5 * (0.02 if big_mean else 1.0)).to(T).to(gpu_device)
    bias = (torch.randn(cout, generator=g) * (40.0 if big_mean else 1.0)).to(gpu_device)
    r = torch.randn(B * H * W, cout, generator=g).to(T).to(gpu_device) if res else None
    geo = ops.conv_gn_partials(T, 1, cin, cout, hin, win, H, W)
    assert geo == ((H // 16) * (W // 16), 256)
    out = torch.empty(B * H * W, cout, device=gpu_device, dtype=T)
    part = torch.full((B, geo[0], cout, 3), float('nan'), device=gpu_device)
    ops.gemm(x, w, out, M=B * H * W, N=cout, K=9 * cin, bias=bias, residual=r, conv=dict(Hin=hin, Win=win, Cin=cin, Hout=H, Wout=W, up=up), gn_part=part)
    plain = torch.empty_like(out)
    ops.gemm(x, w, plain, M=B * H * W, N=cout, K=9 * cin, bias=bias, residual=r, conv=dict(Hin=hin, Win=win, Cin=cin, Hout=H, Wout=W, up=up))
    assert torch.equal(out, plain)                                       # asking for the partials does not change the conv's output
    assert torch.isfinite(part).all()
    # partials per tile: tiles are 16 x 16 pixel blocks in row-major tile order
    y = out.double().cpu().view(B, H // 16, 16, W // 16, 16, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, geo[0], 256, cout)
    pc = part.double().cpu()
    piv = pc[..., 2].unsqueeze(2)
    assert ((y - piv).abs().amin(dim=2) == 0).all()                      # the pivot is one of the tile's own stored values
    S, Q = (y - piv).sum(2), ((y - piv) ** 2).sum(2)
    assert (pc[..., 0] - S).abs().max() <= 1e-4 * (S.abs().max() + 1) and (pc[..., 1] - Q).abs().max() <= 1e-4 * (Q.abs().max() + 1)
    # the GroupNorm from them
    G = 32
    gw, gb = (1 + 0.1 * torch.randn(cout, generator=g)).to(gpu_device), (0.1 * torch.randn(cout, generator=g)).to(gpu_device)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, H * W, cout), device=gpu_device, dtype=torch.uint8)
    for silu in (False, True):
        a = ops.groupnorm_silu_partials(out, gw, gb, torch.empty_like(out), B, H * W, cout, G, 1e-6, silu, part, geo[0], geo[1], ws)
        b_ = ops.groupnorm_silu(out, gw, gb, torch.empty_like(out), B, H * W, cout, G, 1e-6, silu, ws)
        ref = F.group_norm(out.double().cpu().view(B, H * W, cout).permute(0, 2, 1), G, gw.double().cpu(), gb.double().cpu(), 1e-6).permute(0, 2, 1).reshape(B * H * W, cout)
        if silu:
            ref = F.silu(ref)
        assert (a.double().cpu() - ref).abs().max().item() < 3e-2
        d = (a.float() - b_.float()).abs()
        assert (d <= 2.0 ** -7 * (b_.float().abs() + 1)).all() and (d > 0).float().mean().item() < 0.02
    from controlvar_amd._lib import CvarError
    with pytest.raises(CvarError):      # a conv that cannot emit them refuses instead of skipping silently (stride 2)
        ops.gemm(x, w, torch.empty(B * (hin // 2) * (win // 2), cout, device=gpu_device, dtype=T), M=B * (hin // 2) * (win // 2), N=cout, K=9 * cin,
                 conv=dict(Hin=hin, Win=win, Cin=cin, Hout=hin // 2, Wout=win // 2, stride=2), gn_part=part)
    assert ops.conv_gn_partials(T, 2, cin, cout, hin, win, hin // 2, win // 2) is None and ops.conv_gn_partials(torch.float32, 1, cin, cout, hin, win, H, W) is None


def test_vqvae_with_conv_epilogue_statistics_is_as_close_to_fp32_as_without(gpu_device):
    """whole bf16 decoder / encoder with GN_FROM_CONV on and off.  The two differ only by the fp32 summation order of the GroupNorm statistics, but 60 bf16
    layers amplify any one-ulp flip to the mode's own noise level (mean |err| ~ 1e-2 on [-1, 1] pixels) - so the yardstick is the fp32
    parity mode of the same weights: the image with the epilogue statistics must be no farther from it than the image with the stand-alone pass (x 1.25),
    and the two bf16 images no farther from each other than from fp32.  The fp32 mode never takes the halo conv and is untouched."""
    from test_gpu_parity import make_vae as _mk
    from controlvar_amd import models
    vae, vae32 = _mk(160, torch.bfloat16, gpu_device), _mk(160, torch.float32, gpu_device)
    f = torch.randn(2, 32, 16, 16, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    img = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(gpu_device)
    c, fc = vae32.fhat_to_img(f).float(), vae32._encode_f(img)
    try:
        models.VQVAE.GN_FROM_CONV = True
        a, fa = vae.fhat_to_img(f).float(), vae._encode_f(img)
        a2 = vae.fhat_to_img(f).float()
        models.VQVAE.GN_FROM_CONV = False
        b, fb = vae.fhat_to_img(f).float(), vae._encode_f(img)
    finally:
        models.VQVAE.GN_FROM_CONV = True
    assert torch.equal(a, a2)                                            # fixed summation order: bit-reproducible
    e_on, e_off, e_ab = [(u - v).abs().mean().item() for u, v in ((a, c), (b, c), (a, b))]
    print(f'[gn-from-conv] decoder mean |err| vs fp32: with {e_on:.3e}, without {e_off:.3e}; between the two bf16 images {e_ab:.3e}')
    assert e_on <= 1.25 * e_off + 1e-3 and e_ab <= 1.5 * max(e_on, e_off) and e_on < 3e-2
    g_on, g_off = [(u - fc).abs().mean().item() for u in (fa, fb)]
    assert g_on <= 1.25 * g_off + 1e-3 * fc.abs().mean().item()


# ------------------------------------------------------------------------------------------------ round 5: the image conv (conv_c8.hip: 3 channels padded to 8 -> 160-multiples)
@pytest.mark.parametrize('B,H,W,cout,big_mean', [(3, 32, 48, 160, False), (2, 16, 16, 320, False), (1, 64, 32, 160, True)])
def test_conv3x3_image_kernel_against_torch_the_implicit_gemm_and_its_groupnorm_partials(gpu_device, B, H, W, cout, big_mean):
    """conv_c8.hip (what cvar_gemm picks for Cin = 8, Cout % 160 == 0) against torch's fp32 conv2d of the bf16-rounded operands and against the implicit-GEMM tiles
    (tile_cfg 5); NaN images in front of and behind the batch (a tap that leaves its image must read zeros), nothing written behind the output; its GroupNorm partials
    per tile against float64 sums of the values it stored, and the GroupNorm from them against the stand-alone statistics pass."""
    T = torch.bfloat16
    g = torch.Generator().manual_seed(B * 100 + H + cout)
    xs = torch.zeros(B + 2, H, W, 8)
    xs[..., :3] = torch.randn(B + 2, H, W, 3, generator=g)
    xs[0] = float('nan'); xs[-1] = float('nan')
    buf = xs.to(T).to(gpu_device)
    x = buf[1:B + 1].reshape(-1, 8)
    w = torch.zeros(cout, 3, 3, 8)
    w[..., :3] = torch.randn(cout, 3, 3, 3, generator=g) / 27 ** 0.5 * (0.02 if big_mean else 1.0)
    w = w.to(T)
    bias = torch.randn(cout, generator=g) * (40.0 if big_mean else 1.0)
    ref = F.conv2d(x.float().cpu().view(B, H, W, 8).permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, cout)
    geo = ops.conv_gn_partials(T, 1, 8, cout, H, W, H, W)
    assert geo == ((H // 16) * (W // 16), 256)
    outs = {}
    for cfg in (0, 5):
        out = torch.full((B * H * W + 64, cout), float('nan'), device=gpu_device, dtype=T)
        part = torch.full((B, geo[0], cout, 3), float('nan'), device=gpu_device) if cfg == 0 else None
        part0 = part if cfg == 0 else part0
        ops.GEMM_TILE_CFG = cfg
        try:
            ops.gemm(x, w.reshape(cout, 72).to(gpu_device), out, M=B * H * W, N=cout, K=72, bias=bias.to(gpu_device),
                     conv=dict(Hin=H, Win=W, Cin=8, Hout=H, Wout=W), gn_part=part)
        finally:
            ops.GEMM_TILE_CFG = 0
        assert torch.isnan(out[B * H * W:].float()).all()
        outs[cfg] = out[:B * H * W]
        assert torch.isfinite(outs[cfg].float()).all()
        assert close(outs[cfg], ref, T, bf16_rel=1e-2), (cfg, (outs[cfg].float().cpu() - ref).abs().max().item())
        if cfg == 0:
            pc = part.double().cpu()
            assert torch.isfinite(pc).all()
    assert ((outs[0].float() - outs[5].float()).abs().cpu() <= 2.0 ** -7 * (ref.abs() + 1)).all()
    out = outs[0].contiguous()
    y = out.double().cpu().view(B, H // 16, 16, W // 16, 16, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, geo[0], 256, cout)
    piv = pc[..., 2].unsqueeze(2)
    assert ((y - piv).abs().amin(dim=2) == 0).all()
    S, Q = (y - piv).sum(2), ((y - piv) ** 2).sum(2)
    assert (pc[..., 0] - S).abs().max() <= 1e-4 * (S.abs().max() + 1) and (pc[..., 1] - Q).abs().max() <= 1e-4 * (Q.abs().max() + 1)
    gw, gb = (1 + 0.1 * torch.randn(cout, generator=g)).to(gpu_device), (0.1 * torch.randn(cout, generator=g)).to(gpu_device)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, H * W, cout), device=gpu_device, dtype=torch.uint8)
    a = ops.groupnorm_silu_partials(out, gw, gb, torch.empty_like(out), B, H * W, cout, 32, 1e-6, True, part0, geo[0], geo[1], ws)
    b_ = ops.groupnorm_silu(out, gw, gb, torch.empty_like(out), B, H * W, cout, 32, 1e-6, True, ws)
    d = (a.float() - b_.float()).abs()
    assert (d <= 2.0 ** -7 * (b_.float().abs() + 1)).all() and (d > 0).float().mean().item() < 0.02


# ------------------------------------------------------------------------------------------------ round 3: prescaled queries (ABI 14)
@pytest.mark.parametrize('H,Lmax,q_off,l,levels,holes', [
    (2, 1360, 848, 512, None, None),                       # last scale of the pyramid: 4 query blocks, 22 KV tiles
    (3, 400, 110, 72, None, None),                         # partial query block (waves 3 idles), ragged last KV tile
    (2, 310, 182, 128, None, None),
    (1, 70, 60, 2, None, None),                            # two queries
    (2, 300, 0, 300, [2, 10, 28, 60, 110, 182, 300], None),                                    # teacher-forced level mask, whole sequence
    (2, 120, 40, 80, [20, 40, 80, 120], [(0, 0), (0, 0), (20, 40), (40, 80)]),                 # level holes (indep), cached form
])
@pytest.mark.parametrize('scale', [0.125, 1.0])        # 1.0: row 0's scores sit near -180 in the log2 domain - the first tile's shift is < -128 (round-3 NaN: 0 * 2^180)
def test_attention_prescaled_equals_the_reference_softmax(gpu_device, H, Lmax, q_off, l, levels, holes, scale):
    """cvar_attention_prescaled (query rows carry scale * log2 e; the running maximum is subtracted by a fifth k-step on the matrix pipe and
    only moves in bf16 steps) against torch fp64 softmax on the SAME bf16 operands, and against cvar_attention on unscaled queries.  One
    sample's keys carry a spike that forces the maximum to jump late (the rescale branch), another has all scores far below zero."""
    R, C = 3, H * 64
    g = torch.Generator().manual_seed(7)
    kv = torch.randn(R, Lmax, 2 * C, generator=g)
    q = torch.randn(R, l, C, generator=g)
    kv[1, min(Lmax - 1, q_off + l - 1) // 2, :C] *= 12.0                   # a key in the middle of row 1 dominates: m~ jumps at its tile
    q[2] *= 0.05                                                           # near-uniform attention
    q[0] = q[0].abs() * 3.0; kv[0, :, :C] = -kv[0, :, :C].abs()            # every score of row 0 far below zero (the first tile must move m~ DOWN)
    c2 = scale * 1.4426950408889634
    kvd = kv.to(torch.bfloat16).to(gpu_device)
    qd = q.to(torch.bfloat16).to(gpu_device)
    qpd = (q.to(torch.bfloat16).float() * c2).to(torch.bfloat16).to(gpu_device)
    out_p = torch.empty(R * l, C, device=gpu_device, dtype=torch.bfloat16)
    out_u = torch.empty_like(out_p)
    lse = torch.empty(R, H, l, device=gpu_device)
    ops.attention(kvd, out_p, R, H, Lmax, q_off, l, scale, levels, holes=holes, q=qpd.view(R * l, C), prescaled=True, lse=lse)
    ops.attention(kvd, out_u, R, H, Lmax, q_off, l, scale, levels, holes=holes, q=qd.view(R * l, C))
    # reference on the operands the prescaled kernel saw: s_log2 = q' . k
    kf = kvd.double().cpu()[:, :, :C].view(R, Lmax, H, 64).permute(0, 2, 1, 3)
    vf = kvd.double().cpu()[:, :, C:].view(R, Lmax, H, 64).permute(0, 2, 1, 3)
    qf = qpd.double().cpu().view(R, l, H, 64).permute(0, 2, 1, 3)
    s2 = qf @ kf.transpose(-1, -2)                                            # (R, H, l, Lmax), log2 domain
    pos = torch.arange(q_off, q_off + l)
    keys = torch.arange(Lmax)
    if levels:
        ends = torch.tensor(levels)
        lv = torch.searchsorted(ends, pos, right=True)
        vis = keys[None, :] < ends[lv][:, None]
        if holes:
            hl = torch.tensor(holes)[lv]
            vis &= ~((keys[None, :] >= hl[:, :1]) & (keys[None, :] < hl[:, 1:]))
    else:
        vis = (keys[None, :] < q_off + l).expand(l, Lmax)
    s2 = s2.masked_fill(~vis, float('-inf'))
    pr = torch.softmax(s2 * math.log(2.0), dim=-1)
    ref = (pr @ vf).permute(0, 2, 1, 3).reshape(R * l, C)
    err = (out_p.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert torch.isfinite(out_p.float()).all() and err < (1.2e-2 if scale < 1 else 3e-2), err      # bf16 P and bf16 output; scale 1: near-one-hot rows
    ref_lse = torch.logsumexp(s2 * math.log(2.0), dim=-1)
    assert (lse.double().cpu() - ref_lse).abs().max() < 2e-2
    # the two kernels differ only by the rounding of q * c2: same function
    if scale < 1:                                                             # (at |logit| ~ 100 the re-rounded query moves near-ties: compared with the exact softmax above only)
        assert (out_p.float() - out_u.float()).abs().max().item() < 4e-2 * out_u.float().abs().max().item()


@pytest.mark.parametrize('R,l,C,K', [(4, 96, 256, 128),          # 128x128 tiles (generic epilogue)
                                     (2, 9, 768, 768),           # small M: split-K slices + the split-K epilogue kernel
                                     (8, 512, 768, 768)])        # 256x256 tiles (specialised remap epilogue)
def test_gemm_split_alpha_scales_only_the_split_columns(gpu_device, R, l, C, K):
    """cvar_gemm_desc.split_alpha (ABI 14): (acc + bias) * alpha on the columns that go to C_split, one rounding; the arena columns untouched"""
    ops.ensure_splitk_workspace(gpu_device)
    Lmax, off = l + 34, 7
    M, N = R * l, 3 * C
    A, W, b = to_dev(rnd(M, K, seed=1), torch.bfloat16, gpu_device), to_dev(rnd(N, K, seed=2), torch.bfloat16, gpu_device), rnd(N, seed=3).to(gpu_device)
    alpha = 0.03125 * 1.4426950408889634
    outs = []
    for a in (1.0, alpha):
        arena = torch.zeros(R, Lmax, 2 * C, device=gpu_device, dtype=torch.bfloat16)
        qs = torch.zeros(M, C, device=gpu_device, dtype=torch.bfloat16)
        ops.gemm(A, W, arena, M=M, N=N, K=K, bias=b, ldc=2 * C, remap=(l, Lmax, off), split=(qs, C, C), split_alpha=a)
        outs.append((arena, qs))
    assert torch.equal(outs[0][0], outs[1][0])
    acc = A.float() @ W.float().t()[:, :C] + b[:C]
    want = (acc * alpha).to(torch.bfloat16)
    got = outs[1][1]
    assert (got.float() - want.float()).abs().max().item() <= 2 ** -7 * want.float().abs().max().item()        # accumulation order only: <= 1 bf16 ulp of the largest value
    assert (got != want).float().mean().item() < 0.02


# ---- round 4 (second half): the weight-streaming small-M kernel (gemm_skinny.hip) and the row-finishing split-K reduction with the adaLN behind it
def _skinny_case(gpu_device, M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gpu_device)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    return A, W, bias, g


@pytest.mark.parametrize('M,N,K', [(4, 4608, 1536), (16, 1536, 6144), (36, 1536, 1536), (64, 6144, 1536), (100, 4608, 1536), (144, 1536, 1536), (256, 1536, 1536),
                                   (2, 1568, 96), (50, 48, 1536), (33, 4096, 1536), (400, 1536, 1536)])
def test_skinny_gemm_every_epilogue_against_torch_and_the_tile_kernels(gpu_device, M, N, K):
    """tile_cfg 12 sends these bf16 calls to cvar_gemm_skinny_kernel (whole K per workgroup, 8 waves interleaving the k-steps, epilogue in the launch);
    tile_cfg 0 keeps them on the LDS-tiled kernels + split-K.  Same math, other fp32 summation order: both against torch on the bf16 operands, and the
    two against each other inside the rounding of the output type.  Operands sit in NaN-filled arenas (a read outside a tensor cannot hide)."""
    ops.ensure_splitk_workspace(gpu_device)
    A0, W0, bias, g = _skinny_case(gpu_device, M, N, K, M * 13 + N + K)
    # NaN arenas around both operands
    Abuf = torch.full((M + 2, K), float('nan'), device=gpu_device, dtype=torch.bfloat16); Abuf[1:M + 1] = A0; A = Abuf[1:M + 1]
    Wbuf = torch.full((N + 2, K), float('nan'), device=gpu_device, dtype=torch.bfloat16); Wbuf[1:N + 1] = W0; W = Wbuf[1:N + 1]
    l = max(M // 2, 1)
    R = (M + l - 1) // l
    ada = (torch.randn(R, 2 * N, generator=g) * 0.3).to(gpu_device)
    res0 = (torch.randn(M, N, generator=g) * 0.5).to(gpu_device)
    want = A0.float() @ W0.float().t() + bias
    tol = 0.02 * max(1.0, float(want.abs().max()) / 8)

    def run(cfg, kind):
        ops.GEMM_TILE_CFG = cfg
        try:
            if kind == 'gelu':
                out = torch.full((M, N), float('nan'), device=gpu_device, dtype=torch.bfloat16)
                ops.gemm(A, W, out, M=M, N=N, K=K, a_off=0, bias=bias, act=ACT_GELU_TANH)
            elif kind == 'gate_res':
                out = res0.clone()
                ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias, gate=ada, gate_off=N, ldg=2 * N, gate_rows=l, residual=out)
            elif kind == 'f32':
                out = torch.full((M, N), float('nan'), device=gpu_device, dtype=torch.float32)
                ops.gemm(A, W, out, M=M, N=N, K=K, bias=bias)
            else:               # K/V-arena remap + column split with the query factor (the qkv call of inference); N = 3 parts
                Cq = N // 3
                Lmax, off = l + 7, 3
                arena = torch.full((R, Lmax, 2 * Cq), float('nan'), device=gpu_device, dtype=torch.bfloat16)
                qs = torch.full((M, Cq), float('nan'), device=gpu_device, dtype=torch.bfloat16)
                ops.gemm(A, W, arena, M=M, N=N, K=K, bias=bias, ldc=2 * Cq, remap=(l, Lmax, off), split=(qs, Cq, Cq), split_alpha=0.18)
                out = (arena, qs)
            return out
        finally:
            ops.GEMM_TILE_CFG = 0

    kinds = ['gelu', 'gate_res', 'f32'] + (['qkv'] if N % 24 == 0 and M % l == 0 else [])
    for kind in kinds:
        a, b = run(12, kind), run(0, kind)
        if kind == 'qkv':
            Cq = N // 3
            Lmax, off = l + 7, 3
            for got in (a, b):
                arena, qs = got
                assert (qs.float().cpu() - (want[:, :Cq] * 0.18).cpu()).abs().max() < tol
                kv = arena[:, off:off + l].reshape(M, 2 * Cq).float().cpu()
                assert (kv - want[:, Cq:].cpu()).abs().max() < tol
                rest = torch.cat([arena[:, :off].reshape(-1), arena[:, off + l:].reshape(-1)])
                assert torch.isnan(rest.float()).all()                       # rows outside the scale's slots untouched
            continue
        ref = {'gelu': F.gelu(want, approximate='tanh'), 'f32': want,
               'gate_res': res0 + want * ada[:, N:].repeat_interleave(l, 0)[:M]}[kind]
        for got in (a, b):
            assert not torch.isnan(got.float()).any(), kind
            assert (got.float() - ref).abs().max() < tol, (kind, float((got.float() - ref).abs().max()))
        assert (a.float() - b.float()).abs().max() < tol
    # run-to-run determinism of the streaming kernel (fixed wave order in the LDS reduction)
    r0 = run(12, 'f32')
    for _ in range(3):
        assert torch.equal(run(12, 'f32'), r0)


@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K,cfg', [(4, 1536, 6144, 12), (64, 1536, 6144, 12), (144, 1536, 6144, 12), (100, 1536, 1536, 12), (36, 1920, 1920, 12),
                                       (64, 1536, 6144, 0), (256, 1536, 1536, 0), (676, 1536, 6144, 0), (1024, 1536, 1536, 0), (200, 1920, 7680, 0),
                                       (2048, 1536, 1536, 12)])
def test_gemm_with_the_adaln_of_the_next_op_is_bit_identical_to_two_calls(gpu_device, out_dtype, M, N, K, cfg):
    """cvar_gemm_desc.ln_out (ABI 17): proj / fc2 of a small pass also write LN(x) * (1 + scale) + shift of the rows they finish.  Where the call is sliced
    along K the rows are finished by cvar_splitk_rowfin_kernel (slices -> bias / gate / residual -> x -> adaLN in one launch); otherwise the library
    launches cvar_ln_modulate behind the GEMM.  With the slice producer held fixed the result must equal GEMM + cvar_ln_modulate bit for bit: x AND the
    modulated rows.  cfg 0 = LDS-tiled producers (split-K for M <= 1024), cfg 12 = the streaming kernel's K slices / unsliced form; M = 2048: no slices at all."""
    ops.ensure_splitk_workspace(gpu_device)
    A, W, bias, g = _skinny_case(gpu_device, M, N, K, M + 3 * N + K)
    l = max(M // 2, 1)
    R = (M + l - 1) // l
    n_ada = 6 * N
    ada = (torch.randn(R, n_ada, generator=g) * 0.3).to(gpu_device)
    res0 = (torch.randn(M, N, generator=g) * 0.5).to(gpu_device)
    eps = 1e-6
    ops.GEMM_TILE_CFG = cfg
    try:
        x1 = res0.clone()
        u1 = torch.full((M, N), float('nan'), device=gpu_device, dtype=out_dtype)
        ops.gemm(A, W, x1, M=M, N=N, K=K, bias=bias, gate=ada, gate_off=N, ldg=n_ada, gate_rows=l, residual=x1, ln=(u1, ada, 3 * N, 5 * N, n_ada, l, eps))
        # the same call without the request: the slice plan of the streaming kernel depends on whether a row-finishing launch follows, so the unfused
        # reference of cfg 12 is only guaranteed to share the producer when the call is not sliced; compare bits where it does, numbers otherwise
        x2 = res0.clone()
        ops.gemm(A, W, x2, M=M, N=N, K=K, bias=bias, gate=ada, gate_off=N, ldg=n_ada, gate_rows=l, residual=x2)
        u2 = torch.empty(M, N, device=gpu_device, dtype=out_dtype)
        ops.ln_modulate(x1, ada, 3 * N, 5 * N, n_ada, l, u2, M, N, eps)
    finally:
        ops.GEMM_TILE_CFG = 0
    assert not torch.isnan(u1.float()).any()
    # (a) the modulated rows are exactly cvar_ln_modulate of the x this call wrote
    assert torch.equal(u1, u2)
    # (b) x itself: same bits as the unfused call when the producer is the same (always for the tile kernels), same numbers otherwise
    want = res0 + (A.float() @ W.float().t() + bias) * ada[:, N:2 * N].repeat_interleave(l, 0)[:M]
    assert (x1 - want).abs().max() < 0.02 * max(1.0, float(want.abs().max()) / 8)
    if cfg == 0 or M > 1024:
        assert torch.equal(x1, x2)
    else:
        assert (x1 - x2).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('q_off,l,levels', [(848, 512, None), (0, 512, (2, 10, 28, 60, 110, 182, 310, 512))])
def test_attention_64_queries_per_wave_kernel_equals_the_128_query_kernel_bit_for_bit(gpu_device, q_off, l, levels):
    """cvar_attention_prescaled sends scales whose last 256-query workgroup is nearly full AND whose grid still fills the chip (>= 512 workgroups) to
    attn_mfma_bf16_q64_kernel (two 32-query groups per wave share every K / V fragment, tile and barrier); everything else runs the 128-query kernel.  Same
    arithmetic per query: rows 0:2 of an R = 12 call (24 heads x 12 rows x 2 workgroups = 576: the 64-query kernel) equal an R = 2 call (96 workgroups of 256
    queries would not fill the chip: the 128-query kernel) bit for bit, and both match the fp64 softmax."""
    H, Lmax, R = 24, 1360, 12
    C = H * 64
    g = torch.Generator().manual_seed(19)
    kv = (torch.randn(R, Lmax, 2 * C, generator=g) * 0.7).to(torch.bfloat16).to(gpu_device)
    kv[1, (q_off + l) // 2, :C] *= 9.0                       # a dominating key in row 1: the running maximum jumps late
    scale = 0.125
    q = torch.randn(R, l, C, generator=g)
    qp = (q.to(torch.bfloat16).float() * (scale * 1.4426950408889634)).to(torch.bfloat16).to(gpu_device)
    out_big = torch.full((R * l, C), float('nan'), device=gpu_device, dtype=torch.bfloat16)
    lse_big = torch.empty(R, H, l, device=gpu_device)
    ops.attention(kv, out_big, R, H, Lmax, q_off, l, scale, levels, q=qp.view(R * l, C), prescaled=True, lse=lse_big)
    out_small = torch.full((2 * l, C), float('nan'), device=gpu_device, dtype=torch.bfloat16)
    lse_small = torch.empty(2, H, l, device=gpu_device)
    ops.attention(kv[:2].contiguous(), out_small, 2, H, Lmax, q_off, l, scale, levels, q=qp[:2].reshape(2 * l, C).contiguous(), prescaled=True, lse=lse_small)
    assert not torch.isnan(out_big.float()).any()
    assert torch.equal(out_big[:2 * l], out_small) and torch.equal(lse_big[:2], lse_small)
    # and the function itself on rows 0:2 (fp64 softmax of the same bf16 operands, log2 domain)
    kf = kv[:2].double().cpu()[:, :, :C].view(2, Lmax, H, 64).permute(0, 2, 1, 3)
    vf = kv[:2].double().cpu()[:, :, C:].view(2, Lmax, H, 64).permute(0, 2, 1, 3)
    qf = qp[:2].double().cpu().view(2, l, H, 64).permute(0, 2, 1, 3)
    s2 = qf @ kf.transpose(-1, -2)
    pos, keys = torch.arange(q_off, q_off + l), torch.arange(Lmax)
    if levels:
        ends = torch.tensor(levels)
        vis = keys[None, :] < ends[torch.searchsorted(ends, pos, right=True)][:, None]
    else:
        vis = (keys[None, :] < q_off + l).expand(l, Lmax)
    pr = torch.softmax(s2.masked_fill(~vis, float('-inf')) * math.log(2.0), dim=-1)
    ref = (pr @ vf).permute(0, 2, 1, 3).reshape(2 * l, C)
    assert (out_small.double().cpu() - ref).abs().max().item() < 1.2e-2 * ref.abs().max().item()


# ------------------------------------------------------------------------------------------------ round 5: measurement aid of bench.py (ABI 19)
def test_mfma_probe_runs_and_validates_its_arguments(gpu_device):
    """cvar_probe_mfma_bf16 (bench.py's roofline.sustained_peak): refuses short / misaligned operand buffers, and a launch on zeros is not slower than one on
    random operands (the part clocks to its power budget) while both stay below the 2.5 PFLOP/s peak"""
    import bench
    from controlvar_amd import _lib
    lib = _lib.load()
    ops_t = torch.zeros(1 << 17, device=gpu_device, dtype=torch.bfloat16)
    assert lib.cvar_probe_mfma_bf16(None, ops_t.numel() * 2, 10, None, None) == -1
    assert lib.cvar_probe_mfma_bf16(ops_t.data_ptr(), 1024, 10, None, None) == -1
    assert lib.cvar_probe_mfma_bf16(ops_t.data_ptr() + 2, ops_t.numel() * 2 - 2, 10, None, None) == -1
    assert lib.cvar_probe_mfma_bf16(ops_t.data_ptr(), ops_t.numel() * 2, 0, None, None) == -1
    with torch.cuda.device(gpu_device):
        r = bench.sustained_mfma(gpu_device, launches=4, iters=8000)
    assert 300.0 < r['randn'] < 2600.0 and 300.0 < r['zeros'] < 2600.0, r
    assert r['zeros'] > 0.97 * r['randn'], r


# ------------------------------------------------------------------------------------------------ round 5: the 256x192 tile of the partial-round rule
@pytest.mark.parametrize('M,N,K', [(2048, 4608, 1536), (777, 384, 128), (5408, 1536, 6144), (3000, 1920, 1536)])
def test_gemm_256x192_tile_every_hot_epilogue_bit_identical_to_the_256x256_tile(gpu_device, M, N, K):
    """cvar_gemm's 256x192 tile (tile_cfg 27 forces it; the automatic plan takes it where 256x256 tiles fill their last round badly): plain bf16, bias + GELU,
    fp32 gate + residual in place, K/V-arena row remap with the q split - against torch fp32 of the bf16 operands and bit for bit against the forced 256x256 tile
    (same K order per output).  Its wave tile is 64x96: 12 lanes per output row, so four lanes of every wave idle in the row-major epilogue - the case that needs the
    staging stores kept alive (gemm.hip)."""
    T = torch.bfloat16
    A, W, b = rnd(M, K, seed=M), rnd(N, K, seed=N) / K ** 0.5, rnd(N, seed=3)
    Ad, Wd, bd = to_dev(A, T, gpu_device), to_dev(W, T, gpu_device), b.to(gpu_device)
    ref = Ad.float().cpu() @ Wd.float().cpu().t() + b
    gate = rnd((M + 255) // 256, N, seed=5).to(gpu_device)
    res0 = rnd(M, N, seed=6)
    l = 64
    R, L, off = (M + l - 1) // l, 200, 17

    def run(cfg, kind):
        ops.GEMM_TILE_CFG = cfg
        try:
            if kind == 'plain':
                out = torch.full((M + 8, N), float('nan'), device=gpu_device, dtype=T)
                ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=bd)
                assert torch.isnan(out[M:].float()).all()
                return out[:M]
            if kind == 'gelu':
                out = torch.empty(M, N, device=gpu_device, dtype=T)
                ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=bd, act=ACT_GELU_TANH)
                return out
            if kind == 'gate':
                out = res0.to(gpu_device).clone()
                ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=bd, gate=gate, ldg=N, gate_rows=256, residual=out)
                return out
            out = torch.zeros(R * L, N, device=gpu_device, dtype=T)
            ops.gemm(Ad, Wd, out, M=M, N=N, K=K, bias=bd, remap=(l, L, off))
            return out
        finally:
            ops.GEMM_TILE_CFG = 0

    for kind in ('plain', 'gelu', 'gate', 'remap'):
        a, b2 = run(27, kind), run(2, kind)
        assert torch.equal(a, b2), kind
        if kind == 'plain':
            assert close(a, ref, T)
        elif kind == 'gelu':
            assert close(a, F.gelu(ref, approximate='tanh'), T)
        elif kind == 'gate':
            want = res0 + gate.cpu().repeat_interleave(256, 0)[:M] * ref
            assert ((a.cpu() - want).abs() <= 1e-2 * (want.abs() + 1)).all()


# ------------------------------------------------------------------------------------------------ round 6: RPF epilogue of the 256x256 tile
@pytest.mark.parametrize('M,N,K,l,inplace', [(2048, 1536, 1536, 512, True), (4096 + 200, 1536, 6144, 2, True), (2304, 512, 128, 90, False), (2560, 1792, 256, 16, True)])
def test_gemm_gate_residual_lds_prefetch_equals_the_register_form(gpu_device, M, N, K, l, inplace):
    """x += gate * (A W^T + b) (proj / fc2, basic_var.py:208-209) on the eight-wave 256x256 tile: full tiles take the epilogue whose residual rows are prefetched three
    half-passes ahead by LDS DMA (round 6), partial tiles and tile_cfg 28 the register form - the SAME arithmetic in the same order, so the results must be bit-identical;
    rows with ragged M (a partial last row tile), gate rows that change inside a half-pass (l = 2, 90), columns that are not a multiple of 256, out of place and in place."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gpu_device)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(gpu_device)
    b = torch.randn(N, generator=g).to(gpu_device)
    nseq = -(-M // l)
    gate = torch.randn(nseq, 3 * N, generator=g).to(gpu_device)
    x0 = torch.randn(M, N, generator=g).to(gpu_device)
    outs = []
    for cfg in (2, 28):                                        # 2: the eight-wave 256x256 tile wherever it applies (RPF on full tiles); 28: the same without RPF
        old = ops.GEMM_TILE_CFG
        try:
            ops.GEMM_TILE_CFG = cfg
            res = x0.clone()
            out = res if inplace else torch.full_like(x0, float('nan'))
            ops.gemm(A, W, out, M=M, N=N, K=K, bias=b, gate=gate, gate_off=N, ldg=3 * N, gate_rows=l, residual=res, split_k=False)
            torch.cuda.synchronize()
            outs.append(out.clone())
            if not inplace:
                assert torch.equal(res, x0)                    # the residual operand is only read
        finally:
            ops.GEMM_TILE_CFG = old
    assert torch.equal(outs[0], outs[1])
    acc = A.float() @ W.float().t() + b
    ref = x0 + acc * gate[:, N:2 * N].repeat_interleave(l, dim=0)[:M]
    assert (outs[0] - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('kind', ['plain', 'gelu', 'gate', 'f32'])
def test_gemm_streaming_policy_changes_no_bit(gpu_device, kind):
    """round 6: `cvar_gemm` treats outputs of >= 128 MB as streams - non-temporal stores in the specialised epilogues of the 256-row tiles, the LDS-prefetched
    read-modify-write (RPF) for proj / fc2 - and smaller ones as before.  The policy must not change a bit: one call over M rows (above the threshold) against the same
    problem issued as two calls over M / 2 rows each (below it).  The parity fixtures run at small batches and never reach the threshold; this test and the bench do."""
    K = 256
    M, N = (40960, 1536) if kind in ('gate', 'f32') else (40960, 2048)           # fp32: 251 MB / 126 MB per half; bf16: 168 MB / 84 MB per half
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(gpu_device)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(gpu_device)
    b = torch.randn(N, generator=g).to(gpu_device)
    l = 512
    gate = torch.randn(M // l, N, generator=g).to(gpu_device)
    x0 = torch.randn(M, N, generator=g).to(gpu_device) if kind == 'gate' else None
    odt = torch.float32 if kind in ('gate', 'f32') else torch.bfloat16
    assert M * N * (4 if odt == torch.float32 else 2) >= 128 << 20 > (M // 2) * N * (4 if odt == torch.float32 else 2)

    def call(a, out, g_rows):
        m = a.shape[0]
        if kind == 'gate':
            ops.gemm(a, W, out, M=m, N=N, K=K, bias=b, gate=g_rows, ldg=N, gate_rows=l, residual=out)
        else:
            ops.gemm(a, W, out, M=m, N=N, K=K, bias=b, act=ACT_GELU_TANH if kind == 'gelu' else 0)

    whole = x0.clone() if kind == 'gate' else torch.empty(M, N, device=gpu_device, dtype=odt)
    call(A, whole, gate)
    halves = x0.clone() if kind == 'gate' else torch.empty(M, N, device=gpu_device, dtype=odt)
    h = M // 2
    call(A[:h], halves[:h], gate[:h // l])
    call(A[h:], halves[h:], gate[h // l:])
    torch.cuda.synchronize()
    assert torch.equal(whole, halves)
    ref = A[:512].float() @ W.float().t() + b
    if kind == 'gelu':
        ref = F.gelu(ref, approximate='tanh')
    if kind == 'gate':
        ref = x0[:512] + ref * gate[:1]
    assert (whole[:512].float() - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())
