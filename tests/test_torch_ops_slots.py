"""The PyTorch custom-op layer (torch.ops.cvar.*, controlvar_amd/torch_ops.py) and the reference's operator slots
(controlvar_amd/slots.py; models/basic_var.py:15-29: fused_mlp_func, flash_attn_func, slow_attn, memory_efficient_attention,
dropout_add_layer_norm) against plain PyTorch fp32 math on the same inputs, forward and backward.

Tolerances: fp32 mode is the exact-f32 MFMA path -> 2e-4 relative to the output scale; bf16 mode is compared with the fp32 result of the
bf16-rounded inputs -> 2e-2 (bf16 has 8 mantissa bits; intermediate activations are rounded once more)."""
import math

import pytest
import torch
import torch.nn.functional as F


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-6)).item()


TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2}


# ------------------------------------------------------------------------------------------------------------------ host-side (no GPU)
def test_ops_are_registered_with_schemas_and_fail_loudly_on_cpu():
    import controlvar_amd
    ns = controlvar_amd.register_torch_ops()
    from controlvar_amd import torch_ops
    assert len(torch_ops.OPS) >= 15
    for name in torch_ops.OPS:
        op = getattr(ns, name)
        assert op.default._schema.name == 'cvar::' + name
    with pytest.raises(RuntimeError, match='no CPU'):
        ns.linear(torch.randn(4, 8), torch.randn(16, 8))
    with pytest.raises(RuntimeError, match='no CPU'):
        ns.attention(torch.randn(1, 4, 3 * 64), 1, 0, 4, 0.1, [], False)
    from controlvar_amd import slots
    with pytest.raises(RuntimeError, match='GPU only'):
        slots.fused_mlp_func(torch.randn(2, 8), torch.randn(16, 8), torch.randn(8, 16))


def test_fake_kernels_give_shapes_without_a_gpu():
    import controlvar_amd
    ns = controlvar_amd.register_torch_ops()
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a = torch.empty(3, 5, 64, dtype=torch.bfloat16, device='cuda')
        w = torch.empty(128, 64, dtype=torch.bfloat16, device='cuda')
        assert ns.linear(a, w).shape == (3, 5, 128) and ns.linear(a, w, None, 0, None, 1, None, torch.float32).dtype == torch.float32
        o, lse = ns.attention(torch.empty(2, 10, 3 * 2 * 64, device='cuda', dtype=torch.bfloat16), 2, 4, 6, 0.1, [], False)
        assert o.shape == (12, 128) and lse.shape == (2, 2, 6) and lse.dtype == torch.float32
        o2 = ns.attention_kv(torch.empty(2, 10, 2 * 2 * 64, device='cuda', dtype=torch.bfloat16), torch.empty(2, 6, 128, device='cuda', dtype=torch.bfloat16),
                             2, 4, 0.1, [], False)
        assert o2.shape == (12, 128)
        y = ns.ln_modulate(torch.empty(6, 64, device='cuda'), torch.empty(2, 64, device='cuda'), torch.empty(2, 64, device='cuda'), 3, 1e-6, torch.bfloat16)
        assert y.shape == (6, 64) and y.dtype == torch.bfloat16
        assert ns.cfg_sample(torch.empty(4, 7, 4096, device='cuda'), 2, 2, [2.0, -1.0], 900, 0.96, 1, 0, 1).shape == (2, 7)


def test_install_assigns_the_reference_slot_names():
    import types
    from controlvar_amd import slots
    mod = types.SimpleNamespace(fused_mlp_func=None, dropout_add_layer_norm=None, flash_attn_func=None, slow_attn=None, memory_efficient_attention=None)
    slots.install(mod)
    assert mod.fused_mlp_func is slots.fused_mlp_func and mod.flash_attn_func is slots.flash_attn_func
    assert mod.slow_attn is slots.slow_attn and mod.dropout_add_layer_norm is slots.dropout_add_layer_norm
    assert mod.memory_efficient_attention is None                  # xformers slot is left alone (flash takes precedence upstream, basic_var.py:111-115)


# ------------------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_op_forward_backward(gpu_device, dtype):
    ns = torch.ops.cvar
    import controlvar_amd
    controlvar_amd.register_torch_ops()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 37, 256, generator=g)).to(dtype).to(gpu_device).requires_grad_(True)
    w = (torch.randn(512, 256, generator=g) / 16).to(dtype).to(gpu_device).requires_grad_(True)
    b = torch.randn(512, generator=g).to(gpu_device).requires_grad_(True)
    for act in (0, 1):
        y = ns.linear(x, w, b, act)
        ref = F.linear(x.detach().float(), w.detach().float(), b.detach())
        ref = F.gelu(ref, approximate='tanh') if act else ref
        assert y.shape == (3, 37, 512) and y.dtype == dtype
        assert rel_err(y, ref) < TOL[dtype]
        dy = torch.randn(3, 37, 512, generator=g).to(gpu_device)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy.to(dtype))
        xr, wr, br = x.detach().float().requires_grad_(True), w.detach().float().requires_grad_(True), b.detach().clone().requires_grad_(True)
        yr = F.linear(xr, wr, br)
        yr = F.gelu(yr, approximate='tanh') if act else yr
        rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), dy.to(dtype).float())
        assert rel_err(gx, rx) < TOL[dtype] and rel_err(gw, rw) < TOL[dtype] and rel_err(gb, rb) < TOL[dtype]
        assert gx.dtype == dtype and gw.dtype == dtype
    # gate + residual epilogue (x + gamma * f, basic_var.py:208-209)
    gate = torch.randn(3, 512, generator=g).to(gpu_device)
    res = torch.randn(3, 37, 512, generator=g).to(gpu_device)
    y = ns.linear(x.detach(), w.detach(), b.detach(), 0, gate, 37, res, torch.float32)
    ref = res + gate[:, None, :] * F.linear(x.detach().float(), w.detach().float(), b.detach())
    assert y.dtype == torch.float32 and rel_err(y, ref) < TOL[dtype]
    with pytest.raises(TypeError):
        ns.linear(x.detach().half(), w.detach().half())


@gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_mlp_slot(gpu_device, dtype):
    from controlvar_amd import slots
    g = torch.Generator().manual_seed(1)
    C = 128
    x = torch.randn(2, 50, C, generator=g).to(dtype).to(gpu_device).requires_grad_(True)
    w1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).to(dtype).to(gpu_device).requires_grad_(True)
    w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(dtype).to(gpu_device).requires_grad_(True)
    b1 = torch.randn(4 * C, generator=g).to(dtype).to(gpu_device).requires_grad_(True)
    b2 = torch.randn(C, generator=g).to(dtype).to(gpu_device).requires_grad_(True)
    y = slots.fused_mlp_func(x=x, weight1=w1, weight2=w2, bias1=b1, bias2=b2, activation='gelu_approx', save_pre_act=True, return_residual=False,
                             checkpoint_lvl=0, heuristic=0, process_group=None)                 # the exact call of FFN.forward (basic_var.py:44-49)
    f = [t.detach().float().requires_grad_(True) for t in (x, w1, w2, b1, b2)]
    ref = F.linear(F.gelu(F.linear(f[0], f[1], f[3]), approximate='tanh'), f[2], f[4])
    assert rel_err(y, ref) < TOL[dtype]
    dy = torch.randn(2, 50, C, generator=g).to(gpu_device)
    got = torch.autograd.grad(y, (x, w1, w2, b1, b2), dy.to(dtype))
    want = torch.autograd.grad(ref, f, dy.to(dtype).float())
    for a, b_, n in zip(got, want, 'x w1 w2 b1 b2'.split()):
        assert rel_err(a, b_) < 2 * TOL[dtype], n
    with pytest.raises(NotImplementedError):
        slots.fused_mlp_func(x, w1, w2, b1, b2, activation='relu')


def _level_mask(ends, L, dev):
    lvl = torch.zeros(L, dtype=torch.long)
    b = 0
    for k, e in enumerate(ends):
        lvl[b:e] = k
        b = e
    return torch.where(lvl.view(-1, 1) >= lvl.view(1, -1), 0., -torch.inf).view(1, 1, L, L).to(dev)


@gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_slots_forward_and_backward(gpu_device, dtype):
    """slow_attn (B H L c + additive level mask, basic_var.py:117), flash_attn_func (B L H c, basic_var.py:113) incl. the KV-cache
    form (fewer queries than keys), and their gradients, against torch's softmax attention in fp32."""
    from controlvar_amd import slots
    g = torch.Generator().manual_seed(2)
    B, H, L, c = 2, 3, 90, 64
    scale = 0.03125
    mk = lambda *s: (torch.randn(*s, generator=g) * 1.5).to(dtype).to(gpu_device)
    q, k, v = mk(B, H, L, c).requires_grad_(True), mk(B, H, L, c).requires_grad_(True), mk(B, H, L, c).requires_grad_(True)
    ends = [2, 10, 28, 60, 90]
    mask = _level_mask(ends, L, gpu_device)
    o = slots.slow_attn(query=q, key=k, value=v, scale=scale, attn_mask=mask, dropout_p=0.0)
    fq, fk, fv = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax(fq @ fk.transpose(-1, -2) * scale + mask, dim=-1) @ fv
    assert o.shape == (B, H, L, c) and rel_err(o, ref) < TOL[dtype]
    do = torch.randn(B, H, L, c, generator=g).to(gpu_device)
    got = torch.autograd.grad(o, (q, k, v), do.to(dtype))
    want = torch.autograd.grad(ref, (fq, fk, fv), do.to(dtype).float())
    for a, b_, n in zip(got, want, 'qkv'):
        assert rel_err(a, b_) < 3 * TOL[dtype], n
    # flash layout, no mask, KV cache: 32 new queries over 90 keys
    q2 = mk(B, 32, H, c)
    kk, vv = k.detach().transpose(1, 2).contiguous(), v.detach().transpose(1, 2).contiguous()
    o2 = slots.flash_attn_func(q2, kk, vv, dropout_p=0.0, softmax_scale=scale)
    ref2 = torch.softmax(q2.float().transpose(1, 2) @ kk.float().transpose(1, 2).transpose(-1, -2) * scale, dim=-1) @ vv.float().transpose(1, 2)
    assert o2.shape == (B, 32, H, c) and rel_err(o2.transpose(1, 2), ref2) < TOL[dtype]
    o3 = slots.memory_efficient_attention(q2, kk, vv, attn_bias=None, p=0.0, scale=scale)
    assert torch.equal(o2, o3)
    # masks the kernels cannot express are refused, not approximated
    bad = torch.zeros(1, 1, L, L, device=gpu_device)
    bad[..., 5, 3] = -torch.inf
    with pytest.raises(NotImplementedError):
        slots.slow_attn(q, k, v, scale=scale, attn_mask=bad)
    with pytest.raises(NotImplementedError):
        slots.slow_attn(q, k, v, scale=scale, dropout_p=0.1)


@gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_add_layer_norm_slot(gpu_device, dtype):
    """the call of SABlock.fused_forward_wo_cond (basic_var.py:163-171): prenorm, fp32 residual, layer scale, per-token rowscale"""
    from controlvar_amd import slots
    g = torch.Generator().manual_seed(3)
    B, L, C = 2, 33, 128
    x0 = torch.randn(B, L, C, generator=g).to(dtype).to(gpu_device).requires_grad_(True)
    res = torch.randn(B, L, C, generator=g).to(gpu_device).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu_device).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).to(gpu_device).requires_grad_(True)
    ls = (0.5 + torch.rand(C, generator=g)).to(gpu_device).requires_grad_(True)
    rowscale = ((torch.rand(B, 1, generator=g) < 0.7).float() / 0.7).expand(B, L).contiguous().to(gpu_device)
    y, r = slots.dropout_add_layer_norm(x0=x0, residual=res, weight=w, bias=b, dropout_p=0.0, epsilon=1e-6, rowscale=rowscale, layerscale=ls,
                                        prenorm=True, residual_in_fp32=True)
    f = [t.detach().float().requires_grad_(True) for t in (x0, res, w, b, ls)]
    rr = f[1] + f[0] * rowscale[..., None] * f[4]
    yr = F.layer_norm(rr, (C,), f[2], f[3], 1e-6)
    assert y.dtype == dtype and r.dtype == torch.float32
    assert rel_err(r, rr) < 1e-5 and rel_err(y, yr) < TOL[dtype]
    dy, dr = torch.randn(B, L, C, generator=g).to(gpu_device), torch.randn(B, L, C, generator=g).to(gpu_device)
    got = torch.autograd.grad((y, r), (x0, res, w, b, ls), (dy.to(dtype), dr))
    want = torch.autograd.grad((yr, rr), f, (dy.to(dtype).float(), dr))
    for a, b_, n in zip(got, want, 'x0 residual weight bias layerscale'.split()):
        assert rel_err(a, b_) < 3 * TOL[dtype], n
    # first block: no residual yet, no scales (basic_var.py:162)
    y1, r1 = slots.dropout_add_layer_norm(x0.detach(), None, w.detach(), b.detach(), 0.0, 1e-6, prenorm=True, residual_in_fp32=True)
    assert rel_err(r1, x0.detach().float()) < 1e-6 and rel_err(y1, F.layer_norm(x0.detach().float(), (C,), w.detach(), b.detach(), 1e-6)) < TOL[dtype]


@gpu
def test_misc_ops_against_torch(gpu_device):
    ns = torch.ops.cvar
    import controlvar_amd
    controlvar_amd.register_torch_ops()
    g = torch.Generator().manual_seed(4)
    # adaLN modulate (basic_var.py:208) with per-sample rows
    x = torch.randn(3, 20, 192, generator=g).to(gpu_device)
    sc, sh = torch.randn(3, 192, generator=g).to(gpu_device) * 0.2, torch.randn(3, 192, generator=g).to(gpu_device)
    y = ns.ln_modulate(x, sc, sh, 20, 1e-6, torch.float32)
    ref = F.layer_norm(x, (192,), eps=1e-6) * (1 + sc[:, None]) + sh[:, None]
    assert rel_err(y, ref) < 1e-5
    # fused cross-entropy forward + gradient (train_control_var_hpu.py:135,231)
    logits = torch.randn(40, 4096, generator=g).to(gpu_device) * 3
    tg = torch.randint(0, 4096, (40,), generator=g).to(gpu_device)
    loss, dl = ns.ce_fwd_bwd(logits, tg, None, 1.0 / 40, torch.float32)
    lr = logits.clone().requires_grad_(True)
    lref = F.cross_entropy(lr, tg, reduction='none')
    assert rel_err(loss, lref) < 1e-5
    assert rel_err(dl, torch.autograd.grad(lref.mean(), lr)[0]) < 1e-4
    # GroupNorm + SiLU over NHWC (vae_modules.py:18-19,58)
    B, HW, C = 2, 64, 64
    xn = torch.randn(B * HW, C, generator=g).to(gpu_device)
    w, b = (1 + 0.1 * torch.randn(C, generator=g)).to(gpu_device), (0.1 * torch.randn(C, generator=g)).to(gpu_device)
    yn = ns.groupnorm_silu(xn, w, b, B, HW, 32, 1e-6, True)
    refn = F.silu(F.group_norm(xn.view(B, HW, C).permute(0, 2, 1), 32, w, b, 1e-6)).permute(0, 2, 1).reshape(B * HW, C)
    assert rel_err(yn, refn) < 1e-4
    # sampler: greedy == argmax of the CFG-combined logits (control_var.py:501-505)
    lg = torch.randn(4, 6, 4096, generator=g).to(gpu_device)
    idx = ns.cfg_sample(lg, 2, 2, [3.0, -2.0], 1, 0.0, 0, 0, 1)
    assert torch.equal(idx.long(), (3.0 * lg[:2] - 2.0 * lg[2:]).argmax(-1))
    # AdamW step of one tensor against torch.optim.AdamW
    p = torch.randn(1000, generator=g).to(gpu_device)
    gr = torch.randn(1000, generator=g).to(gpu_device)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    pt.grad = gr.clone()
    opt.step()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ns.adamw_(p, gr, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.05, 1, 1.0)
    assert rel_err(p, pt.detach()) < 1e-6
    # inference form: K/V arena + the call's queries == the packed-arena op on the same numbers
    qkv = torch.randn(2, 50, 3 * 128, generator=g).to(gpu_device, torch.bfloat16)
    o_packed, _ = ns.attention(qkv, 2, 30, 20, 0.125, [], False)
    o_kv = ns.attention_kv(qkv[:, :, 128:].contiguous(), qkv[:, 30:50, :128].contiguous(), 2, 30, 0.125, [], False)
    assert torch.equal(o_packed, o_kv)
    with pytest.raises(ValueError):
        ns.attention_kv(qkv, qkv[:, 30:50, :128].contiguous(), 2, 30, 0.125, [], False)                # a packed arena is not a K/V arena
    # the bf16 inference kernel proper: queries pre-multiplied by scale * log2(e) (one rounding more than o_kv's operands)
    qpre = (qkv[:, 30:50, :128].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16).contiguous()
    o_pre = ns.attention_kv_prescaled(qkv[:, :, 128:].contiguous(), qpre, 2, 30, [])
    assert o_pre.shape == o_kv.shape and rel_err(o_pre, o_kv) < 3e-2
    with pytest.raises(TypeError):
        ns.attention_kv_prescaled(qkv[:, :, 128:].float().contiguous(), qpre.float(), 2, 30, [])
    # a bad status from the C ABI surfaces as RuntimeError
    with pytest.raises(RuntimeError):
        ns.attention(torch.zeros(1, 4, 3 * 64, device=gpu_device, dtype=torch.bfloat16), 1, 2, 5, 0.1, [], False)      # q_off + l > Lmax
