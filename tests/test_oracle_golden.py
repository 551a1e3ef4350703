"""Pins oracle/ against the fixtures recorded from the reference (tests/golden/make_golden.py).

These are CPU tests: they prove the restatement reproduces the reference's outputs on the
same seeded weights/inputs, so that the GPU parity tests may use the oracle as the checker.
"""
import numpy as np
import pytest
import torch

from conftest import golden
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VaeConfig, VarConfig, phi_index_map
from controlvar_amd.synth import synth_images, synth_vae_state, synth_var_state
from oracle import var_ref, vqvae_ref
from oracle.interp import area_matrix, bicubic_matrix
from oracle.vqvae_ref import MSQuant

torch.set_num_threads(8)


def t(a):
    return torch.from_numpy(np.asarray(a))


def split_ids(ids, mf=1):
    out, o = [], 0
    for p in PN:
        n = mf * p * p
        out.append(t(ids[:, o:o + n]).long())
        o += n
    return out


@pytest.fixture(scope='module')
def vae32():
    sd = synth_vae_state(VaeConfig(ch=32))
    return sd, MSQuant(sd, PN, phi_index_map(10))


def test_interp_matrices_match_torch_fixture():
    g = golden('interp')
    f = t(g['f']).double()
    for p in PN[:-1]:
        A = t(area_matrix(16, p))
        got = torch.einsum('ih,bchw,jw->bcij', A, f, A)
        assert np.abs(got.numpy() - g[f'area_{p}']).max() < 2e-6
        M = t(bicubic_matrix(p, 16))
        got = torch.einsum('ih,bchw,jw->bcij', M, t(g[f'h_{p}']).double(), M)
        assert np.abs(got.numpy() - g[f'bicubic_{p}']).max() < 5e-6


def test_bicubic_matrices_between_consecutive_scales_match_torch():
    """the low-resolution path resizes f_hat between consecutive scales (1->2, 2->3, ... 13->16, quant.py:176): the dense matrices of both
    table builders (oracle and product host code) against F.interpolate(mode='bicubic') live"""
    import torch.nn.functional as F
    from controlvar_amd.pyramid import bicubic_matrix as product_matrix
    g = torch.Generator().manual_seed(3)
    for a, b in zip(PN[:-1], PN[1:]):
        x = torch.randn(2, 3, a, a, generator=g, dtype=torch.float64)
        want = F.interpolate(x, size=(b, b), mode='bicubic')
        for build in (bicubic_matrix, product_matrix):
            M = t(np.asarray(build(a, b), dtype=np.float64))
            got = torch.einsum('ih,bchw,jw->bcij', M, x, M)
            assert (got - want).abs().max() < 1e-10, (a, b)


def test_phi_map():
    assert phi_index_map(10) == [0, 0, 1, 1, 1, 2, 2, 3, 3, 3]          # SURVEY A15, measured on the reference


def test_next_input_all_scales(vae32):
    sd, msq = vae32
    g = golden('next_input')
    for si, p in enumerate(PN):
        f2, nxt = msq.next_input(si, t(g[f'fhat_in_{si}']), t(g[f'h_{si}']))
        assert (f2 - t(g[f'fhat_out_{si}'])).abs().max() < 2e-5
        assert (nxt - t(g[f'next_{si}'])).abs().max() < 2e-5


def _check_tokenizer(tag, ch):
    g = golden(f'tokenizer_{tag}')
    sd = synth_vae_state(VaeConfig(ch=ch))
    msq = MSQuant(sd, PN, phi_index_map(10))
    img = synth_images(int(g['nimg']), 256, seed=1)
    with torch.no_grad():
        f = vqvae_ref.img_to_f(sd, img)
    assert (f - t(g['f'])).abs().max() < 2e-4 * max(1.0, float(np.abs(g['f']).max()))
    # integer path on the REFERENCE's f: ids must be identical
    ids, margins = msq.f_to_idx(t(g['f']), return_margins=True)
    ids = torch.cat(ids, dim=1).numpy()
    mism = ids != g['ids'].astype(np.int64)
    mg = torch.cat(margins, dim=1).numpy()
    assert mism.sum() == 0 or mg[mism].max() < 1e-4, f'{mism.sum()} id mismatches, margins {mg[mism]}'
    fh = msq.f_to_idx(t(g['f']), to_fhat=True)
    assert (fh[-1] - t(g['fhat_last'])).abs().max() < 1e-4
    assert (fh[3] - t(g['fhat_s3'])).abs().max() < 1e-4
    gi = split_ids(g['ids'].astype(np.int64))
    var_in = torch.cat(msq.idx_to_var_input(gi), dim=1)
    assert (var_in[:, ::5] - t(g['var_in'])).abs().max() < 2e-5
    with torch.no_grad():
        rec = vqvae_ref.idxBl_to_img(sd, msq, gi)
    assert (rec[:, :, 100:116, 60:76] - t(g['rec_crop'])).abs().max() < 2e-4
    assert (rec[:, :, -20:-4, 200:216] - t(g['rec_crop2'])).abs().max() < 2e-4
    assert (rec.mean(dim=(2, 3)) - t(g['rec_mean'])).abs().max() < 1e-5


def test_tokenizer_tiny():
    _check_tokenizer('ch32', 32)


@pytest.mark.slow
def test_tokenizer_full():
    _check_tokenizer('ch160', 160)


@pytest.mark.parametrize('cos', [False, True])
def test_block(cos):
    g = golden('block_cos' if cos else 'block')
    cfg = VarConfig(depth=30 if cos else 2, embed_dim=128, num_heads=2)
    sd = synth_var_state(cfg, seed=3)
    cond = t(g['cond'])
    ada = var_ref.ada_params(sd, 'blocks.0.', cond, 6, var_ref.FP32)
    cache = var_ref.KVCache(cfg.depth)
    y0 = var_ref.block(sd, 0, cfg, t(g['x0']), ada, cache, None, var_ref.FP32)
    y1 = var_ref.block(sd, 0, cfg, t(g['x1']), ada, cache, None, var_ref.FP32)
    lvl = torch.tensor([0, 0, 1, 1, 1, 1, 1, 1, 1, 1]).view(1, 10, 1)
    bias = torch.where(lvl >= lvl.transpose(1, 2), 0., -torch.inf).reshape(1, 1, 10, 10)
    ym = var_ref.block(sd, 0, cfg, t(g['xm']), ada, None, bias, var_ref.FP32)
    for got, key in ((y0, 'y0'), (y1, 'y1'), (ym, 'ym')):
        assert (got - t(g[key])).abs().max() < 2e-5, key


@pytest.mark.parametrize('tag,mf', [('d2', 2), ('var_d2', 1)])
def test_forward_logits(tag, mf):
    g = golden(f'forward_{tag}')
    cfg = VarConfig(depth=2, mask_factor=mf, control=(mf == 2), multi_cond=(mf == 2))
    sd = synth_var_state(cfg)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    am = logits.argmax(-1).numpy()
    mism = am != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4
    assert (logits.double().sum(-1).float() - t(g['lsum'])).abs().max() < 2e-2


def test_forward_logits_d12_width():
    """BASELINE config 2 anchor: the oracle at d12 width (C=768, 12 heads, 12 blocks) against the reference's logits"""
    g = golden('forward_d12')
    cfg = VarConfig(depth=12)
    sd = synth_var_state(cfg)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 5e-4
    am = logits.argmax(-1).numpy()
    mism = am != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 5e-4


def test_bf16_emulated_d12_trace_is_the_fp32_trace_up_to_rounding():
    """gen_d12_bf16emu.npz (the oracle with bf16 storage points, B=8) is what config 2's GPU test is measured against; its rows 0..1 use
    other labels than the reference's fp32 trace, so the link is structural: ids in range, margins positive, and the recorded logits
    are the logits of the recorded ids' scale (finite, max below the recorded absmax)"""
    g = golden('gen_d12_bf16emu')
    ids, margin = g['ids'].astype(np.int64), g['margin']
    assert ids.shape == (8, 1360) and ids.min() >= 0 and ids.max() < 4096
    assert (margin >= 0).all() and np.isfinite(g['logit_samples']).all()
    o = 0
    for si, p in enumerate((1, 2, 3, 4, 5, 6, 8, 10, 13, 16)):
        l = 2 * p * p
        assert np.abs(g['logit_samples'][:, o:o + l]).max() <= float(g['absmax_per_scale'][si]) + 1e-4
        o += l


def test_bf16_reference_fixtures_pin_the_oracle_in_both_precisions():
    """forward_d12_bf16ref.npz (recorded from the reference: fp32, CPU bf16 autocast) also carries the oracle's bf16 emulation: the oracle
    in fp32 must reproduce the reference's fp32 logits, its emulation must reproduce the recorded emulation, and the recorded ordering -
    the emulation is CLOSER to the reference's fp32 than the reference's own autocast - is what the GPU test leans on"""
    from oracle.vqvae_ref import Prec
    g = golden('forward_d12_bf16ref')
    cfg = VarConfig(depth=12)
    sd = synth_var_state(cfg)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        l32 = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))[:, ::9, ::31]
        lemu = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']), prec=Prec(True))[:, ::9, ::31]
    assert (l32 - t(g['ref_fp32'])).abs().max() < 5e-4
    assert (lemu - t(g['emu'])).abs().max() < 1e-5 * float(g['absmax']) + 1e-4          # same code, same machine class: reproducible
    assert g['d_emu_fp32'][0] < g['d_autocast_fp32'][0] and g['d_emu_fp32'][1] < g['d_autocast_fp32'][1]
    assert g['d_autocast_fp32'][0] > 1e-3 * float(g['absmax'])                           # the reference's own bf16 misses 1e-3 of max|logit| (max norm)


@pytest.mark.slow
def test_oracle_walks_the_references_bf16_trace():
    """gen_d12_bf16ref.npz: rows 0..1 of the reference's autocast trace; the fp32 oracle forced along those ids reproduces the fp32
    reference's logits on that path (the reference was walked along the same ids when the fixture was recorded)"""
    g = golden('gen_d12_bf16ref')
    cfg = VarConfig(depth=12)
    sdv, sd = synth_vae_state(VaeConfig(ch=160)), synth_var_state(cfg)
    ids = [i[:2] for i in split_ids(g['ids'].astype(np.int64), mf=2)]
    trace = {}
    with torch.no_grad():
        var_ref.generate(sd, cfg, MSQuant(sdv, PN, phi_index_map(10)), 2, t(g['labels'])[:2], 4.0, top_k=1, cond_type=t(g['types'])[:2], force_idx=ids, trace=trace)
    lg = torch.cat([x[:, :, 5::128] for x in trace['logits']], dim=1)
    ref = t(g['ref_fp32'])[:2]
    assert (lg - ref).abs().max() < 2e-3 * max(1.0, float(ref.abs().max()))


def _gen_check(name, cfg, B, labels, cfg_scale, cond_type=None, four=False, teach=None, top_k=1, top_p=0.0, seed=0,
               vae_ch=32, img_tol=5e-4, wseed=0, logit_tol=2e-3, mean_tol=1e-4, id_frac=0.0, **kw0):
    g = golden(name)
    sdv = synth_vae_state(VaeConfig(ch=vae_ch))
    msq = MSQuant(sdv, PN, phi_index_map(10))
    sd = synth_var_state(cfg, wseed)
    kw = dict(kw0)
    if teach is not None:
        kw[teach] = split_ids(g['c_ids'].astype(np.int64))
    trace = {}
    with torch.no_grad():
        f_hats = var_ref.generate(sd, cfg, msq, B, labels, cfg_scale, top_k=top_k, top_p=top_p, g_seed=seed,
                                  cond_type=cond_type, four_way=four, trace=trace, **kw)
        img = var_ref.decode_fhat(sdv, f_hats)
    ids = torch.cat(trace['idx'], dim=1).numpy()
    ref_ids = g['ids'].astype(np.int64)
    assert ids.shape == ref_ids.shape
    mism = ids != ref_ids
    assert mism.mean() <= id_frac, f'{name}: {mism.sum()} token mismatches (min ref margin at mismatch {g["margin"][mism].min():.3e})'
    if mism.any():             # (only where id_frac > 0) a flipped draw puts the remaining scales on another trajectory: nothing further to compare
        return
    lg = torch.cat(trace['logits'], dim=1)[:2, :, ::128][:, ::3]
    assert (lg - t(g['logit_samples'])).abs().max() < logit_tol * max(1.0, float(np.abs(g['logit_samples']).max()))
    assert (img[:, :, 100:116, 60:76] - t(g['img_crop'])).abs().max() < img_tol
    assert (img[:, :, -20:-4, 200:216] - t(g['img_crop2'])).abs().max() < img_tol
    assert (img.mean(dim=(2, 3)) - t(g['img_mean'])).abs().max() < mean_tol


def test_generate_d2_b2():
    _gen_check('gen_d2_b2', VarConfig(depth=2), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))


def test_generate_d2_b4_none():
    _gen_check('gen_d2_b4none', VarConfig(depth=2), 4, torch.tensor([1, 10, 100, 999]), 4.0, cond_type=None)


def test_generate_conditional_cmask():
    _gen_check('gen_d2_cmask', VarConfig(depth=2), 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]),
               four=True, teach='c_mask')


def test_generate_conditional_cimg():
    _gen_check('gen_d2_cimg', VarConfig(depth=2), 2, torch.tensor([5, 6]), (3.0, 2.0, 1.0), cond_type=torch.tensor([2, 3]),
               four=True, teach='c_img')


def test_generate_sampled_same_generator():
    """top_k=900/top_p=0.96 with the same CPU generator stream reproduces the reference's draw."""
    _gen_check('gen_d2_b2_sampled', VarConfig(depth=2), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]),
               top_k=900, top_p=0.96, seed=42)


def test_generate_plain_var():
    _gen_check('gen_var_d2_b2', VarConfig(depth=2, mask_factor=1, control=False, multi_cond=False), 2, torch.tensor([3, 7]), 4.0)


def test_generate_cos_attn():
    _gen_check('gen_d30n_b2', VarConfig(depth=30, embed_dim=128, num_heads=2), 2, torch.tensor([3, 7]), 4.0,
               cond_type=torch.tensor([3, 0]))


@pytest.mark.slow
def test_generate_d12_config1():
    """BASELINE.json configs[0]: d12 ControlVAR, B=2, CPU greedy decode."""
    _gen_check('gen_d12_b2', VarConfig(depth=12), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), vae_ch=160)


def test_bf16_emulation_modes_agree():
    """The two query-rounding modes of the bf16 emulation (Prec.q_round: 'prescaled' = the HIP storage point bf16(q * scale * log2 e),
    'plain' = bf16(q) with the scale in fp32) are the same function up to one bf16 rounding of q: their teacher-forced logits differ by
    bf16 noise only, and both differ from the fp32 oracle by the same order.  A wrong constant in either would show as an O(1) gap."""
    from oracle.vqvae_ref import Prec
    cfg = VarConfig(depth=2)
    sd = synth_var_state(cfg)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    labels, types = torch.tensor([3, 900]), torch.tensor([1, 2])
    with torch.no_grad():
        l32 = var_ref.forward_logits(sd, cfg, labels, x, types)
        la = var_ref.forward_logits(sd, cfg, labels, x, types, prec=Prec(True))
        lb = var_ref.forward_logits(sd, cfg, labels, x, types, prec=Prec(True, 'plain'))
    amax = float(l32.abs().max())
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt()) / amax
    assert 0 < rms(la, lb) < 2e-3 and rms(la, l32) < 5e-3 and rms(lb, l32) < 5e-3, (rms(la, lb), rms(la, l32), rms(lb, l32))


@pytest.mark.slow
def test_generate_d24_headline_model():
    """The model the BASELINE metric is quoted on, full width (d24 + ch160 VQVAE), B=2 greedy: the oracle (what bench.py's cpu_baseline
    leg times and what smoke() checks against) reproduces the reference's recorded trace token for token."""
    _gen_check('gen_d24_b2', VarConfig(depth=24), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), vae_ch=160)


def test_sampler_masks():
    g = golden('sampler')
    gen = torch.Generator().manual_seed(77)
    logits = torch.randn(2, 6, 4096, generator=gen) * 3
    for (k, p) in [(900, 0.96), (0, 0.5), (50, 0.0), (1, 0.0)]:
        lg = var_ref.topk_topp_mask_(logits.clone(), k, p)
        kept = np.packbits(torch.isfinite(lg).numpy(), axis=-1)
        assert (kept == g[f'kept_{k}_{p}']).all()
    # greedy == the reference's top_k=1 multinomial
    assert (var_ref.sample(logits.clone(), 1, 0.0, None).numpy() == g['idx_1_0.0']).all()


# ------------------------------------------------------------------------------ shared_aln + type_pos
VARIANT = VarConfig(depth=2, shared_aln=True, type_pos=True)
VARIANT_VAR = VarConfig(depth=2, mask_factor=1, control=False, multi_cond=False, shared_aln=True)


def test_variant_state_tables_match_the_reference_key_order():
    from controlvar_amd.spec import var_state_shapes
    assert list(var_state_shapes(VARIANT)) == [str(k) for k in golden('forward_d2v')['keys']]
    assert list(var_state_shapes(VARIANT_VAR)) == [str(k) for k in golden('gen_var_d2s_b2')['keys']]


def test_variant_forward_logits():
    g = golden('forward_d2v')
    sd = synth_var_state(VARIANT, 5)
    gen = torch.Generator().manual_seed(22)
    x = torch.randn(2, VARIANT.pyramid.L - VARIANT.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, VARIANT, t(g['labels']), x, t(g['types']))
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    mism = logits.argmax(-1).numpy() != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4
    assert (logits.double().sum(-1).float() - t(g['lsum'])).abs().max() < 2e-2


def test_variant_generate():
    """conditional_infer_cfg of the reference ignores type_pos; autoregressive_infer_cfg applies it from scale 1 on"""
    _gen_check('gen_d2v_b2', VARIANT, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=5)
    _gen_check('gen_d2v_cmask', VARIANT, 2, torch.tensor([5, 6]), (4.0, 3.0, 2.0), cond_type=torch.tensor([2, 3]), four=True,
               teach='c_mask', wseed=5)
    _gen_check('gen_var_d2s_b2', VARIANT_VAR, 2, torch.tensor([3, 7]), 4.0, wseed=6)


# ------------------------------------------------------------------------------ SABlock (aln < 0)
SA = VarConfig(depth=2, sa_block=True, layer_scale=0.1)
SA0 = VarConfig(depth=2, sa_block=True)


@pytest.mark.parametrize('tag,cfg,seed', [('d2sa', SA, 7), ('d2sa0', SA0, 8)])
def test_sa_block_forward_logits_and_key_order(tag, cfg, seed):
    from controlvar_amd.spec import var_state_shapes
    g = golden(f'forward_{tag}')
    assert list(var_state_shapes(cfg)) == [str(k) for k in g['keys']]
    sd = synth_var_state(cfg, seed)
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    mism = logits.argmax(-1).numpy() != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4


def test_sa_block_generate():
    _gen_check('gen_d2sa_b2', SA, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=7)


# ------------------------------------------------------------------------------ bidirectional (image first)
BIDI = VarConfig(depth=2, bidirectional=True, type_pos=True)


def test_bidirectional_image_first_forward_and_generate():
    from controlvar_amd.spec import var_state_shapes
    g = golden('forward_d2b')
    assert list(var_state_shapes(BIDI)) == [str(k) for k in g['keys']]
    sd = synth_var_state(BIDI, 9)
    gen = torch.Generator().manual_seed(24)
    x = torch.randn(2, BIDI.pyramid.L - BIDI.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, BIDI, t(g['labels']), x, t(g['types']), mask_first=False)
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    mism = logits.argmax(-1).numpy() != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4
    _gen_check('gen_d2b_b2', BIDI, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=9, mask_first=False)


# ---- separate_decoding / indep masks, the two-pass inference branch, more_smooth
SEPDEC = {'d2s': (VarConfig(depth=2, separate_decoding=True), 11), 'd2si': (VarConfig(depth=2, separate_decoding=True, indep=True), 12)}


@pytest.mark.parametrize('tag', list(SEPDEC))
def test_separate_decoding_forward_logits_and_level_tables(tag):
    """masked teacher-forced logits of the separate_decoding models against the reference; and the (level end, hole) description the
    attention kernels use reproduces the reference's attn_bias_for_masking exactly"""
    from controlvar_amd.spec import attention_bias_matrix, attention_levels, var_state_shapes
    cfg, seed = SEPDEC[tag]
    g = golden(f'forward_{tag}')
    sd = synth_var_state(cfg, seed)
    assert list(var_state_shapes(cfg)) == [str(k) for k in g['keys']]
    gen = torch.Generator().manual_seed(25)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    mism = logits.argmax(-1).numpy() != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4
    ends, holes = attention_levels(cfg)
    L = cfg.pyramid.L
    vis = np.zeros((L, L), bool)
    for p in range(L):
        k = next(i for i, e in enumerate(ends) if p < e)
        vis[p, :ends[k]] = True
        if holes and holes[k][1] > holes[k][0]:
            vis[p, holes[k][0]:holes[k][1]] = False
    assert (vis == attention_bias_matrix(cfg)).all() and (vis == (sd['attn_bias_for_masking'][0, 0] == 0).numpy()).all()
    assert len(ends) == 20 and (holes is None) == (tag == 'd2s')


def test_separate_decoding_two_pass_generate():
    _gen_check('gen_d2s_b2', SEPDEC['d2s'][0], 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=11)


def test_separate_decoding_indep_generate_and_conditional():
    _gen_check('gen_d2si_b2', SEPDEC['d2si'][0], 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=12)
    _gen_check('gen_d2si_cmask', SEPDEC['d2si'][0], 2, torch.tensor([5, 6]), (4.0, 3.0, 2.0), cond_type=torch.tensor([2, 3]), four=True, teach='c_mask', wseed=12)


def test_more_smooth_reproduces_the_reference_draws():
    """more_smooth with the same CPU generator stream: id draw (multinomial over the in-place masked logits) then the Gumbel noise.
    Every drawn id must equal the reference's (they depend on the generator stream and on the logits only through the kept set); the
    soft embeddings are softmax((logits + g) / tau) with tau down to 0.0135 at the last scale, which amplifies fp32 summation-order
    differences ~75x, so later-scale logits and pixels are compared at 1 % / 2e-2 instead of the 2e-3 / 5e-4 of the hard path."""
    loose = dict(logit_tol=1e-2, img_tol=2e-2, mean_tol=1e-3)
    _gen_check('gen_d2_smooth', VarConfig(depth=2), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=900, top_p=0.96, seed=42, more_smooth=True, **loose)
    _gen_check('gen_d2_smooth_greedy', VarConfig(depth=2), 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=1, seed=1, more_smooth=True)
    _gen_check('gen_d2_smooth_cmask', VarConfig(depth=2), 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]), four=True, teach='c_mask',
               top_k=900, top_p=0.96, seed=7, more_smooth=True, id_frac=0.01, **loose)      # 3-term CFG on soft inputs: a few draws flip
    _gen_check('gen_d2s_smooth', SEPDEC['d2s'][0], 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=900, top_p=0.96, seed=42, more_smooth=True, wseed=11, **loose)


# ---- separator (recorded with the special_embed index shim of make_golden.make_cvar)
SEPARATOR = {'d2p': (VarConfig(depth=2, separator=True), 13), 'd2psi': (VarConfig(depth=2, separator=True, separate_decoding=True, indep=True), 14)}


@pytest.mark.parametrize('tag', list(SEPARATOR))
def test_separator_state_layout_forward_and_generate(tag):
    from controlvar_amd.spec import var_state_shapes
    cfg, seed = SEPARATOR[tag]
    g = golden(f'forward_{tag}')
    assert list(var_state_shapes(cfg)) == [str(k) for k in g['keys']]
    sd = synth_var_state(cfg, seed)
    py = cfg.pyramid
    assert py.L == 1378 and sd['head.weight'].shape[0] == 4096 + 18 and sd['special_embed.weight'].shape[0] == 18
    gen = torch.Generator().manual_seed(26)
    x = torch.randn(2, len(py.code_positions()) - py.first_l, 32, generator=gen)
    with torch.no_grad():
        logits = var_ref.forward_logits(sd, cfg, t(g['labels']), x, t(g['types']))
    assert logits.shape == (2, 1378, 4114)
    assert (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max() < 1e-4
    mism = logits.argmax(-1).numpy() != g['argmax'].astype(np.int64)
    assert mism.sum() == 0 or g['margin'][mism].max() < 1e-4
    _gen_check(f'gen_{tag}_b2', cfg, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), wseed=seed)


def test_lowres_reconstruction_matches_the_reference(vae32):
    """embed_to_fhat(all_to_max_scale=False) / idxBl_to_img(same_shape=False) (quant.py:171-180, vqvae.py:91-104): the oracle's per-scale
    f_hat list and the images decoded from it against the reference's own (lowres.npz)"""
    sd, msq = vae32
    g = golden('lowres')
    ids = split_ids(g['ids'])
    ms_h = [msq.embed(i, p) for i, p in zip(ids, PN)]
    low = vqvae_ref.embed_to_fhat_lowres(msq, ms_h)
    for si, pn in enumerate(PN):
        assert low[si].shape == (2, 32, pn, pn)
        assert (low[si] - t(g[f'low_{si}'])).abs().max() < 2e-5, si
    imgs = vqvae_ref.idxBl_to_img_lowres(sd, msq, ids)
    for si, pn in enumerate(PN):
        assert imgs[si].shape == (2, 3, 16 * pn, 16 * pn)
        assert (imgs[si][:, :, :16, :16] - t(g[f'img_crop_{si}'])).abs().max() < 2e-3, si
        assert (imgs[si].mean(dim=(2, 3)) - t(g[f'img_mean_{si}'])).abs().max() < 1e-4, si
