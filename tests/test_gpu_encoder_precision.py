"""The tokenizer's middle precision: a bf16 model whose ENCODER runs in split bf16 ("bf16x3": every operand as hi + lo bf16, three MFMA
products per multiply, fp32 accumulate, fp32 activations) or on the exact-f32 MFMA, in front of the always-exact quantizer (quant.py:196-213).  What is at stake is
the id agreement with the reference's fp32 encoder (vae_modules.py:144-160; train_control_var_hpu.py:157-176 takes the training labels from it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden, ids_parity, record  # noqa: E402
from controlvar_amd import models, ops  # noqa: E402
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN  # noqa: E402
from controlvar_amd.synth import synth_images  # noqa: E402

F32, BF16 = torch.float32, torch.bfloat16


def t(a):
    return torch.from_numpy(np.asarray(a))


def _split_ref(y):
    hi = y.to(BF16)
    lo = (y - hi.float()).to(BF16)
    return torch.cat((hi, lo, hi), dim=1)


@pytest.mark.parametrize('M,C,Cpad', [(1000, 160, 480), (4097, 4, 16), (333, 640, 1920), (7, 32, 104)])
def test_split3_is_hi_lo_hi(gpu_device, M, C, Cpad):
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, C, generator=g) * torch.logspace(-3, 3, M).unsqueeze(1)).to(gpu_device)
    out = torch.full((M, Cpad), 7.0, device=gpu_device, dtype=BF16)
    ops.split3(x, out, M, C, Cpad)
    assert torch.equal(out[:, :3 * C], _split_ref(x))
    assert not out[:, 3 * C:].any()
    hi, lo = out[:, :C].double(), out[:, C:2 * C].double()
    rel = ((hi + lo - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -16, rel                           # two bf16 roundings: 2^-9 * 2^-9 (round to nearest: half of that each)


@pytest.mark.parametrize('silu', [True, False])
def test_groupnorm_split3_equals_split_of_the_fp32_groupnorm(gpu_device, silu):
    B, HW, C = 3, 32 * 32, 320
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B * HW, C, generator=g) * 2 + 0.7).to(gpu_device)
    w, b = torch.randn(C, generator=g).to(gpu_device), torch.randn(C, generator=g).to(gpu_device)
    ws = torch.empty(ops.groupnorm_ws_bytes(B, HW, C), device=gpu_device, dtype=torch.uint8)
    y = ops.groupnorm_silu(x, w, b, torch.empty_like(x), B, HW, C, 32, 1e-6, silu, ws)
    out = torch.empty(B * HW, 3 * C, device=gpu_device, dtype=BF16)
    ops.groupnorm_silu_split3(x, w, b, out, B, HW, C, 32, 1e-6, silu, ws)
    assert torch.equal(out, _split_ref(y))


def test_split_product_is_far_closer_to_fp32_than_bf16(gpu_device):
    """one 3x3 conv of the encoder's shape three ways: exact-f32 MFMA, plain bf16, split bf16 through the SAME bf16 kernel with K = 3 x 9 Cin"""
    vae = models.build_vae(ch=32, compute_dtype=BF16, encoder_precision='bf16x3').to(gpu_device)
    vae32 = models.build_vae(ch=32, compute_dtype=F32).to(gpu_device)
    vaeb = models.build_vae(ch=32, compute_dtype=BF16).to(gpu_device)
    B, H, C = 2, 64, 32
    x = torch.randn(B * H * H, C, generator=torch.Generator().manual_seed(0)).to(gpu_device)
    name = 'encoder.down.0.block.0.conv1'
    ref, _, _ = vae32._conv(x, name, B, H, H)
    x3, _, _ = vae._hp_conv(x, name, B, H, H)
    pb, _, _ = vaeb._conv(x.to(BF16), name, B, H, H, out_dtype=F32)
    sc = ref.abs().max().item()
    e3, eb = (x3 - ref).abs().max().item() / sc, (pb - ref).abs().max().item() / sc
    print(f'[x3] conv 32->32 at 64x64: split bf16 {e3:.2e}, plain bf16 {eb:.2e} of max|y|')
    assert e3 < 2e-5 and e3 < eb / 50


@pytest.mark.parametrize('prec', ['fp32', 'bf16x3'])
def test_encoder_precisions_against_the_reference_ids(gpu_device, prec):
    """ch160 tokenizer of a bf16 model on the reference's fixture images (tokenizer_ch160.npz: ids of the reference's fp32 encoder): 'fp32' must reproduce them
    strictly; 'bf16x3' is measured and must agree on >= 99 % (plain bf16: 0.647, profiles/r05c_parity_report.json)"""
    g = golden('tokenizer_ch160')
    vae = models.build_vae(ch=160, compute_dtype=BF16, encoder_precision=prec).to(gpu_device)
    img = synth_images(int(g['nimg']), 256, seed=1).to(gpu_device)
    f = vae._encode_f(img).cpu()
    err_f = float((f - t(g['f'])).abs().max())
    ids = torch.cat(vae.img_to_idxBl(img), dim=1).cpu().numpy()
    ref = g['ids'].astype(np.int64)
    agree = float((ids == ref).mean())
    print(f'[parity] encoder_precision={prec}: max|df| {err_f:.3e} (max|f| {float(np.abs(g["f"]).max()):.2f}), id agreement with the reference {agree:.4f} of {ref.size}')
    record(f'img_to_idxBl ch160 bf16 model, encoder {prec} vs reference fp32 ids', kind='ids', flips=int((ids != ref).sum()), total=int(ref.size), agreement=agree, err_f=err_f,
           strict=prec == 'fp32', tol=0.0)
    if prec == 'fp32':
        ids_parity(ids, ref, np.zeros(ref.shape, np.float32), 0.0, 'bf16 model with the fp32 encoder', strict=True)
        assert err_f < 2e-4
    else:
        assert err_f < 2e-3
        assert agree >= 0.99, agree


def test_encoder_precisions_on_a_larger_sample_and_what_flips_cost_downstream(gpu_device):
    """64 synthetic images: ids of every encoder precision against the fp32 parity mode (whose ids ARE the reference's: 0 flips over every fixture), and what the
    flips mean downstream - PSNR of idxBl_to_img(ids) against idxBl_to_img(ids of the fp32 mode), both through the same fp32 decoder."""
    n = 64
    img = synth_images(n, 256, seed=11).to(gpu_device)
    v32 = models.build_vae(ch=160, compute_dtype=F32).to(gpu_device)
    ids_ref = v32.img_to_idxBl(img)
    rec_ref = v32.idxBl_to_img(ids_ref, same_shape=True, last_one=True)
    cat_ref = torch.cat(ids_ref, dim=1)
    orig_mse = float(((rec_ref - img) ** 2).mean())
    out = {}
    for prec in ('bf16', 'bf16x3', 'fp32'):
        v = models.build_vae(ch=160, compute_dtype=BF16, encoder_precision=prec).to(gpu_device)
        ids = v.img_to_idxBl(img)
        cat = torch.cat(ids, dim=1)
        agree = float((cat == cat_ref).float().mean())
        per_img = (cat == cat_ref).float().mean(dim=1)
        rec = v32.idxBl_to_img(ids, same_shape=True, last_one=True)
        mse = float(((rec - rec_ref) ** 2).mean())
        psnr = float('inf') if mse == 0 else 10 * np.log10(4.0 / mse)               # pixel range [-1, 1]
        mse_img = float(((rec - img) ** 2).mean())
        out[prec] = dict(agreement=agree, images_identical=int((per_img == 1).sum()), psnr_vs_reference_ids_db=psnr, recon_mse=mse_img)
        print(f'[x3] encoder {prec}: id agreement {agree:.4f} ({out[prec]["images_identical"]} of {n} images identical), reconstruction from these ids vs from the reference ids: '
              f'PSNR {psnr:.1f} dB; MSE against the input image {mse_img:.5f} (reference ids: {orig_mse:.5f})')
        del v
    record('encoder precisions on 64 synthetic images vs the fp32 mode', kind='encoder_precision', **{k: v for k, v in out.items()}, recon_mse_reference_ids=orig_mse)
    assert out['fp32']['agreement'] == 1.0
    assert out['bf16x3']['agreement'] >= 0.99 and out['bf16x3']['agreement'] > out['bf16']['agreement']


def test_conv_with_a_misaligned_bias_falls_back_to_the_statistics_pass(gpu_device):
    """VQVAE._conv asks for GroupNorm partials by shape alone; the kernel that emits them also needs a 16-byte aligned bias.  A state dict assigned from a flat
    buffer can hand over a 4-byte aligned one: the conv must then run WITHOUT partials (the GroupNorm behind it makes its own statistics pass) instead of raising, and
    the encode must still run and give the features of the aligned parameters up to bf16 noise."""
    vae = models.build_vae(ch=160, compute_dtype=BF16).to(gpu_device)         # 160 -> 160 at 256 x 256: a conv that emits partials
    img = synth_images(2, 256, seed=4).to(gpu_device)
    want = vae._encode_f(img).clone()
    P = vae._pack()
    name = 'encoder.down.0.block.0.conv1'
    b = P['conv'][name]['b']
    flat = torch.zeros(b.numel() + 1, device=gpu_device, dtype=torch.float32)
    flat[1:] = b
    P['conv'][name]['b'] = flat[1:]                                  # same values, 4 bytes off a 16-byte boundary
    assert P['conv'][name]['b'].data_ptr() % 16 == 4
    x = torch.randn(2 * 256 * 256, 160, device=gpu_device).to(BF16)
    ok, _, _ = vae._conv(x, 'encoder.down.0.block.1.conv1', 2, 256, 256)
    assert getattr(ok, '_gn_part', None) is not None                 # the aligned sibling does emit them
    out, _, _ = vae._conv(x, name, 2, 256, 256)
    assert getattr(out, '_gn_part', None) is None                    # fell back: no partials ride on the tensor
    got = vae._encode_f(img)
    # (the statistics pass sums in another order than the conv's partials: the GroupNorm coefficients move by ~1 ulp and bf16 roundings behind it flip - the features
    #  agree to bf16 noise, which is what the ids of a bf16 encoder are sensitive to anyway: 0.83 id agreement between the two routes on this input)
    assert float((got - want).abs().max()) < 0.05 * float(want.abs().max())


@pytest.mark.parametrize('with_res', [False, True])
def test_halo_conv_fp32_output_form_equals_the_implicit_gemm_tiles(gpu_device, with_res):
    """round 6: the LDS-halo 3x3 kernel with an fp32 output (+ fp32 residual) - the form the split-bf16 encoder's convs take (480 = 3 x 160 split channels -> 160) -
    against the implicit-GEMM tiles on the same operands (tile_cfg 5 keeps a conv there; the two sum K in different orders: chunk-major vs tap-major) and against torch."""
    import torch.nn.functional as F
    B, H, Cin, Cout = 2, 32, 480, 320
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B * H * H, Cin, generator=g).to(BF16).to(gpu_device)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).to(BF16).to(gpu_device)
    b = torch.randn(Cout, generator=g).to(gpu_device)
    res = torch.randn(B * H * H, Cout, generator=g).to(gpu_device) if with_res else None
    outs = []
    for cfg in (0, 5):
        old = ops.GEMM_TILE_CFG
        try:
            ops.GEMM_TILE_CFG = cfg
            out = torch.full((B * H * H + 3, Cout), float('nan'), device=gpu_device)
            ops.gemm(x, w, out, M=B * H * H, N=Cout, K=9 * Cin, bias=b, residual=res, conv=dict(Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, up=0))
            torch.cuda.synchronize()
            assert torch.isnan(out[B * H * H:]).all()
            outs.append(out[:B * H * H].clone())
        finally:
            ops.GEMM_TILE_CFG = old
    xr = x.float().view(B, H, H, Cin).permute(0, 3, 1, 2)
    wr = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, b, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout)
    if with_res:
        ref = ref + res
    sc = float(ref.abs().max())
    assert float((outs[0] - ref).abs().max()) < 2e-5 * sc + 1e-4
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5 * sc + 1e-4
