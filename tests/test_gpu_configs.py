"""BASELINE.json configs 2, 3 and 4 on the MI355X, each anchored to a fixture recorded in the build container
(tests/golden/make_golden.py: the reference itself for fp32, the pinned oracle with bf16 storage points for bf16) and
completed by size-independent properties where the oracle cannot follow (B = 64, sampling).

  config 2  d12 autoregressive_infer_cfg, bf16, B in {1, 8, 64}, greedy + the reference's sampling defaults (top_k=900, top_p=0.96)
  config 3  d24 training step (tokenise -> forward -> CE -> backward -> clip -> AdamW), fp32 against the reference, bf16 properties
  config 4  d30 (cos-attention) at FULL width, B=4 / cond_type=None (all four condition types), and conditional_infer_cfg cfg=(4,4,4)

bf16 bound.  north_star asks for "within 1e-3 on bf16 logits".  Round 3 pins the mode to the REFERENCE's own bf16: the reference runs
under torch.autocast('cpu', bfloat16) in the build container, and its logits differ from its own fp32 logits by 9.7e-3 (max) / 1.8e-3
(RMS) of max|logit| at d12 - no bf16 pipeline, the reference's included, is within 1e-3 of fp32 in the max norm.  The tests
`*_among_the_references_bf16` / `*_along_the_references_bf16_trace` therefore assert that the HIP path is no farther from the reference's fp32
logits than the reference's autocast is (max and RMS), and within 1e-3 RMS of the oracle's bf16 emulation; the older test below keeps the
max-norm number against the emulation (2x the measured value, BF16_REL_BOUND)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden, ids_parity, maxabs_on, record, rows_ok_per_sample  # noqa: E402
from controlvar_amd import models  # noqa: E402
from controlvar_amd import train as T  # noqa: E402
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VarConfig  # noqa: E402
from controlvar_amd.synth import synth_images  # noqa: E402

F32, BF16 = torch.float32, torch.bfloat16
# measured on MI355X (round 2): worst per-scale max|logit_hip - logit_oracle_bf16emu| / max|logit| of the d12 model over B = 1, 8, 64
BF16_REL_MEASURED = 4.1e-3        # B=1: 3.2e-3, B=8: 4.0e-3, B=64: 3.7e-3
BF16_REL_BOUND = 2 * BF16_REL_MEASURED


def t(a):
    return torch.from_numpy(np.asarray(a))


def split(ids, mf=2):
    out, o = [], 0
    for p in PN:
        out.append(ids[:, o:o + mf * p * p])
        o += mf * p * p
    return out


def build(depth, dtype, dev, ch=160):
    vae = models.build_vae(ch=ch, compute_dtype=dtype).to(dev)
    m = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True, compute_dtype=dtype, cond_drop_rate=0.0).to(dev).eval()
    return vae, m


check_ids = ids_parity            # (flips, rows_ok); strict (zero flips) unless a call says strict=False - see conftest.ids_parity


# ---------------------------------------------------------------------------------------------------------------- config 2
def test_forward_d12_fp32_matches_reference(gpu_device):
    """d12-width teacher-forced logits (C=768, 12 heads, 12 blocks) against the reference's own (forward_d12.npz)"""
    g = golden('forward_d12')
    vae, m = build(12, F32, gpu_device, ch=32)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, 1358, 32, generator=gen).to(gpu_device)
    with torch.no_grad():
        logits = m(t(g['labels']), x, t(g['types']), True).cpu()
    err = (logits[:, ::9, ::31] - t(g['logits_sample'])).abs().max().item()
    print(f'd12 fp32 forward: max |logit - reference| = {err:.2e} (max |logit| {float(g["absmax"]):.1f})')
    assert err < 3e-3
    check_ids(logits.argmax(-1), g['argmax'], g['margin'], 3e-3, 'd12 forward argmax')
    assert (logits.double().sum(-1).float() - t(g['lsum'])).abs().max() < 0.5


@pytest.mark.parametrize('B', [1, 8, 64])
def test_generate_d12_bf16_config2(gpu_device, B):
    """bf16 d12 generation, teacher-forced with the oracle's ids on the rows the fixture covers (labels arange(B), types
    arange(B) % 4: the first min(B, 8) rows are the fixture's rows): per-scale CFG logits within BF16_REL_BOUND of the bf16-emulating
    oracle, greedy ids equal wherever the oracle's top-1 margin exceeds 2 x BF16_REL_MEASURED x max|logit| of the scale (two logits, each off by at
    most the round-2 measured error: a constant, not this run's own error); rows ride in batches of 1, 8
    and 64 (different GEMM tilings / split-K partitions) with the same bound."""
    g = golden('gen_d12_bf16emu')
    vae, m = build(12, BF16, gpu_device)
    labels, types = torch.arange(B) % 1000, torch.arange(B) % 4
    nref = min(B, 8)
    ref_ids = split(t(g['ids']).long())
    kw = dict(g_seed=0, cfg=4.0, top_k=1, cond_type=types, _trace=True)
    if B > nref:                                        # rows the fixture does not cover are forced with the model's own greedy ids
        m.autoregressive_infer_cfg(B, labels, **kw)
        own = [x.long().cpu() for x in m.last_trace['idx']]
        forced = [torch.cat((r[:nref], o[nref:]), dim=0) for r, o in zip(ref_ids, own)]
    else:
        forced = [r[:nref] for r in ref_ids]
    img = m.autoregressive_infer_cfg(B, labels, _force_idx=forced, **kw)
    tr = m.last_trace
    assert img.shape == (B, 3, 512, 256) and torch.isfinite(img).all()
    samples = t(g['logit_samples'])                      # rows 0..3, every position, vocabulary entries 5::128
    margin = t(g['margin'])
    worst_rel, o = 0.0, 0
    nrow = min(nref, samples.shape[0])
    for si, p in enumerate(PN):
        l = 2 * p * p
        got = tr['logits'][si][:nrow, :, 5::128].cpu()
        ref = samples[:nrow, o:o + l]
        err = (got - ref).abs().max().item()
        rel = err / float(g['absmax_per_scale'][si])
        worst_rel = max(worst_rel, rel)
        check_ids(tr['idx'][si][:nref].cpu(), ref_ids[si][:nref], margin[:nref, o:o + l].numpy(), 2 * BF16_REL_MEASURED * float(g['absmax_per_scale'][si]), f'd12 bf16 vs emulation B={B} scale {si}', strict=False)
        o += l
    print(f'd12 bf16 B={B}: worst per-scale logit error relative to max|logit| = {worst_rel:.3e} (bound {BF16_REL_BOUND:.1e}; north_star 1e-3)')
    assert worst_rel < BF16_REL_BOUND


def test_generate_d12_bf16_sampling_defaults(gpu_device):
    """config 2, sampled mode (top_k=900, top_p=0.96, seed 42 - train_control_var_hpu.py:338): the draw uses a different generator than
    torch.multinomial, so the checks are the filter's contract: reproducible per seed, every drawn id lies inside the top-900 of its own
    CFG-combined logits, and the draw is not the greedy decode."""
    vae, m = build(12, BF16, gpu_device)
    B = 8
    labels, types = torch.arange(B) % 1000, torch.arange(B) % 4
    kw = dict(cfg=4.0, top_k=900, top_p=0.96, cond_type=types, _trace=True)
    a = m.autoregressive_infer_cfg(B, labels, g_seed=42, **kw)
    tr = m.last_trace
    ids_a = torch.cat(tr['idx'], dim=1).cpu()
    for si in range(len(PN)):
        lg, ids = tr['logits'][si], tr['idx'][si].long()
        chosen = lg.gather(-1, ids.unsqueeze(-1))
        rank = (lg > chosen).sum(-1)
        assert int(rank.max()) < 900, (si, int(rank.max()))
    b = m.autoregressive_infer_cfg(B, labels, g_seed=42, **kw)
    assert torch.equal(a, b) and torch.equal(ids_a, torch.cat(m.last_trace['idx'], dim=1).cpu())
    m.autoregressive_infer_cfg(B, labels, g_seed=43, **kw)
    assert not torch.equal(ids_a, torch.cat(m.last_trace['idx'], dim=1).cpu())
    m.autoregressive_infer_cfg(B, labels, g_seed=42, cfg=4.0, top_k=1, cond_type=types, _trace=True)
    greedy = torch.cat(m.last_trace['idx'], dim=1).cpu()
    assert (greedy != ids_a).float().mean() > 0.2
    assert 0 <= int(ids_a.min()) and int(ids_a.max()) < 4096 and a.shape == (B, 3, 512, 256)


# ------------------------------------------------------------------------------------ config 2: bf16 pinned to the REFERENCE's bf16
def _d(a, b):
    d = (a.double() - b.double())
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def _fourway(what, hip, ref32, refac, emu, amax):
    """the four distances VERDICT r2 asked for, as (max, RMS) relative to max|logit|: HIP-bf16 vs the reference in fp32, the reference's
    own bf16 autocast vs its fp32, HIP-bf16 vs the reference's autocast, HIP-bf16 vs this repo's bf16 emulation"""
    out = {k: [v / amax for v in _d(*pair)] for k, pair in dict(hip_vs_ref_fp32=(hip, ref32), ref_autocast_vs_ref_fp32=(refac, ref32),
                                                               hip_vs_ref_autocast=(hip, refac), hip_vs_emulation=(hip, emu),
                                                               emulation_vs_ref_fp32=(emu, ref32)).items()}
    print(f'[bf16] {what} (max, RMS relative to max|logit| = {amax:.2f}): ' + '; '.join(f'{k} {v[0]:.2e} / {v[1]:.2e}' for k, v in out.items()))
    record(what, kind='bf16_logits', absmax=amax, **out)
    return out


def test_forward_d12_bf16_among_the_references_bf16(gpu_device):
    """north_star: "within 1e-3 on bf16 logits".  The yardstick is the reference itself: forward_d12_bf16ref.npz holds the d12-width
    teacher-forced logits of the reference in fp32 AND under torch.autocast('cpu', bfloat16) (control_var.py:568-651), plus the oracle's
    bf16 emulation, on one input.  Asserted: the HIP bf16 path is no farther from the reference's fp32 logits than the reference's own
    bf16 autocast is (x1.1, max and RMS), and its RMS distance to the emulation (same storage points, different accumulation order) is
    below 1e-3 * max|logit|.  All four distances are printed and recorded (profiles/r03_parity_report.json)."""
    g = golden('forward_d12_bf16ref')
    vae, m = build(12, BF16, gpu_device, ch=32)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, 1358, 32, generator=gen).to(gpu_device)
    with torch.no_grad():
        hip = m(t(g['labels']), x, t(g['types']), True).float().cpu()[:, ::9, ::31]
    r = _fourway('forward_d12 bf16', hip, t(g['ref_fp32']), t(g['ref_autocast']), t(g['emu']), float(g['absmax']))
    assert r['hip_vs_ref_fp32'][0] <= 1.1 * r['ref_autocast_vs_ref_fp32'][0]
    assert r['hip_vs_ref_fp32'][1] <= 1.1 * r['ref_autocast_vs_ref_fp32'][1]
    assert r['hip_vs_emulation'][1] <= 1e-3
    # second, independent emulation mode (queries rounded BEFORE the scale, as the reference's autocast does): the HIP logits must sit as
    # close to it - a mistake in the prescale constant cannot hide in an oracle that mirrors the implementation
    from oracle import var_ref
    from oracle.vqvae_ref import Prec
    from controlvar_amd.synth import synth_var_state
    cfg = VarConfig(depth=12)
    with torch.no_grad():
        plain = var_ref.forward_logits(synth_var_state(cfg), cfg, t(g['labels']), x.cpu(), t(g['types']), prec=Prec(True, 'plain'))[:, ::9, ::31]
    dmax, drms = [v / float(g['absmax']) for v in _d(hip, plain)]
    print(f'[bf16] forward_d12 HIP vs the plain-rounding emulation: {dmax:.2e} / {drms:.2e}')
    record('forward_d12 bf16 vs plain-rounding emulation', kind='bf16_logits', absmax=float(g['absmax']), hip_vs_emulation_plain=[dmax, drms])
    assert drms <= 1.2e-3 and dmax <= 2 * max(r['hip_vs_emulation'][0], BF16_REL_MEASURED)


def test_generate_d12_bf16_along_the_references_bf16_trace(gpu_device):
    """config 2 on the reference's own bf16 trace (gen_d12_bf16ref.npz: autoregressive_infer_cfg under CPU bf16 autocast, B=8, greedy; the
    fp32 reference and the bf16 emulation walked along the same ids): the HIP bf16 model is forced along those ids and its per-scale
    CFG-combined logits are placed among the three.  Asserted per scale as in the forward test; greedy ids agree with the reference's
    autocast ids wherever the autocast margin exceeds the distance between the two bf16 implementations."""
    g = golden('gen_d12_bf16ref')
    vae, m = build(12, BF16, gpu_device)
    B = 8
    labels, types = t(g['labels']), t(g['types'])
    ids = split(t(g['ids']).long())
    m.autoregressive_infer_cfg(B, labels, g_seed=0, cfg=4.0, top_k=1, cond_type=types, _force_idx=ids, _trace=True)
    tr = m.last_trace
    o, flips, emu_sq = 0, 0, 0.0
    for si, p in enumerate(PN):
        l = 2 * p * p
        hip = tr['logits'][si][:4, :, 5::128].float().cpu()
        sl = slice(o, o + l)
        amax = float(g['absmax_per_scale'][si])
        r = _fourway(f'gen_d12 bf16 scale {si}', hip, t(g['ref_fp32'])[:, sl], t(g['ref_autocast'])[:, sl], t(g['emu'])[:, sl], amax)
        assert r['hip_vs_ref_fp32'][0] <= 1.1 * r['ref_autocast_vs_ref_fp32'][0], si
        assert r['hip_vs_ref_fp32'][1] <= 1.1 * r['ref_autocast_vs_ref_fp32'][1], si
        # RMS distance of two bf16 pipelines with the same storage points and different fp32 summation orders: a per-scale sample of it moves by ~10 %
        # whenever a kernel's summation order changes (0.73e-3 ... 1.02e-3 over the ten scales); the 1e-3 of north_star is asserted on the whole
        # generation below, a single scale gets the sample spread on top
        assert r['hip_vs_emulation'][1] <= 1.25e-3, si
        emu_sq += (r['hip_vs_emulation'][1] ** 2) * l
        n, _ = check_ids(tr['idx'][si].cpu(), ids[si], g['margin_autocast'][:, sl], 2.5 * r['hip_vs_ref_autocast'][0] * amax,
                         f'gen_d12 bf16 greedy ids vs the reference autocast trace, scale {si}', strict=False)
        flips += n
        o += l
    emu_rms = (emu_sq / o) ** 0.5
    record('gen_d12 bf16 all scales vs emulation', kind='bf16_logits', absmax=0.0, hip_vs_emulation=[0.0, emu_rms])
    assert emu_rms <= 1e-3, emu_rms
    print(f'd12 bf16 along the reference autocast trace: {flips} of {B * 1360} greedy ids differ (all below the margin bound); RMS distance to the emulation over the generation {emu_rms:.2e}')


def test_one_sample_across_batch_sizes_and_gemm_plans(gpu_device):
    """with the small-M GEMM plans (tile_cfg 12: streaming kernel, slice counts, three-stage tiles) the fp32 summation order of a transformer
    GEMM depends on M, so a sample's logits differ slightly between the B = 1 / 8 / 32 graphs and between SMALL_M_KERNEL on and off.  Pinned here: the
    same sample (same label / type / greedy) is generated at B = 1 with the small-M kernel; every other (batch, plan) combination is FORCED along its ids and
    must give (i) CFG-combined logits within 2 x the measured bf16 distance to the emulation (a bound of the mode, not of this run) and (ii) the same greedy
    id wherever the B = 1 run's top-1 margin exceeds that distance.  Rows of one batch must be bit-identical to each other."""
    from controlvar_amd import ops as O
    vae, m = build(12, BF16, gpu_device)
    lab, typ = 17, 2

    def run(B, force=None):
        m.autoregressive_infer_cfg(B, torch.full((B,), lab), g_seed=5, cfg=4.0, top_k=1, cond_type=torch.full((B,), typ), _trace=True,
                                   **({'_force_idx': [f.expand(B, -1).contiguous() for f in force]} if force is not None else {}))
        tr = m.last_trace
        return [x.clone() for x in tr['idx']], [x.float().clone() for x in tr['logits']]

    try:
        O.SMALL_M_KERNEL = True
        ids1, lg1 = run(1)
        worst, flips = 0.0, 0
        for small in (True, False):
            O.SMALL_M_KERNEL = small
            for B in (1, 8, 32):
                if small and B == 1:
                    continue
                ids, lg = run(B, force=[i[:1] for i in ids1])
                for si in range(len(lg)):
                    a, b = lg[si], lg1[si]
                    assert torch.equal(a[:1].expand_as(a), a), (small, B, si)                      # every row of the batch carries the same sample: same bits
                    amax = float(b.abs().max())
                    d = float((a[:1] - b[:1]).abs().max()) / amax
                    worst = max(worst, d)
                    assert d <= BF16_REL_BOUND, (small, B, si, d)
                    t2 = b[:1].topk(2, dim=-1).values
                    margin = (t2[..., 0] - t2[..., 1])
                    mism = a[:1].argmax(-1) != b[:1].argmax(-1)
                    flips += int(mism.sum())
                    assert not bool((mism & (margin > BF16_REL_BOUND * amax)).any()), (small, B, si)
        print(f'[bf16] one sample across B = 1 / 8 / 32 and both GEMM plans: largest logit distance {worst:.2e} of max|logit|, {flips} argmax flips (all below the margin bound)')
        record('d12 bf16 one sample across batch sizes / GEMM plans', kind='bf16_logits', worst_rel=worst, flips=flips)
    finally:
        O.SMALL_M_KERNEL = True


def test_deterministic_plan_one_sample_is_bit_identical_at_any_batch_size(gpu_device):
    """VERDICT r5 next #7a: ``deterministic_plan=True`` keeps every transformer GEMM on the unsliced tile kernels, so a row's fp32 summation order does not depend on the
    batch it rides in: the same (label, condition type, g_seed) must give the SAME logits bit for bit and the same greedy AND sampled tokens at B = 1 / 3 / 8 / 32 (every
    row carries the sample; row 0 is compared across batch sizes), and the decoded images of row 0 must be equal too.  The default plan is the test above."""
    vae, m = build(12, BF16, gpu_device)
    m.deterministic_plan = True
    lab, typ = 17, 2
    try:
        for kw in (dict(top_k=1), dict(top_k=900, top_p=0.96)):
            base = None
            for B in (1, 3, 8, 32):
                img = m.autoregressive_infer_cfg(B, torch.full((B,), lab), g_seed=5, cfg=4.0, cond_type=torch.full((B,), typ), _trace=True, **kw)
                tr = m.last_trace
                ids = torch.cat([x[:1] for x in tr['idx']], dim=1).clone()
                lgs = [x[:1].float().clone() for x in tr['logits']]
                if kw['top_k'] == 1:
                    for x in tr['logits']:
                        assert torch.equal(x[:1].expand_as(x), x)                 # greedy: every row of the batch is the same sample
                if base is None:
                    base = (ids, lgs, img[:1].clone())
                    continue
                assert torch.equal(ids, base[0]), (kw, B, int((ids != base[0]).sum()))
                for si, (a, b) in enumerate(zip(lgs, base[1])):
                    assert torch.equal(a, b), (kw, B, si, float((a - b).abs().max()))
                assert torch.equal(img[:1], base[2]), (kw, B)
    finally:
        m.deterministic_plan = False


# ---------------------------------------------------------------------------------------------------------------- config 4
def _gen_check(m, g, B, labels, scale, types, what, four=False, c_mask=None, tol=3e-3):
    if four:
        img = m.conditional_infer_cfg(B, labels, g_seed=0, cfg=scale, top_k=1, cond_type=types, c_mask=c_mask, _trace=True)
    else:
        img = m.autoregressive_infer_cfg(B, labels, g_seed=0, cfg=scale, top_k=1, cond_type=types, _trace=True)
    ids = torch.cat(m.last_trace['idx'], dim=1).cpu()
    nm, ok = check_ids(ids[:t(g['ids']).shape[0]], g['ids'], g['margin'], tol, what)
    ok = rows_ok_per_sample(ok, img.shape[0])
    img = img.cpu()
    assert maxabs_on(img[:, :, 100:116, 60:76] - t(g['img_crop']), ok) < 5e-3
    assert maxabs_on(img.mean(dim=(2, 3)) - t(g['img_mean']), ok) < 5e-4
    return img


def test_d30_full_width_fp32_matches_reference(gpu_device):
    """d30 (cos-attention with learned temperature, C=1920, 30 heads, 30 blocks) + the full VQVAE, fp32 mode, token for token against the
    reference's trace: B=4 with cond_type=None (-> condition types [0,1,2,3], control_var.py:387-389) and conditional_infer_cfg with
    cfg=(4,4,4) and c_mask."""
    vae, m = build(30, F32, gpu_device)
    g = golden('gen_d30_b4none')
    _gen_check(m, g, 4, torch.tensor([1, 10, 100, 999]), 4.0, None, 'd30 B=4 cond_type=None')
    g2 = golden('gen_d30_cmask')
    c_ids = split(t(g2['c_ids']).long(), mf=1)
    _gen_check(m, g2, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), torch.tensor([2, 3]), 'd30 conditional_infer_cfg', four=True, c_mask=c_ids)


def test_d24_full_width_fp32_matches_reference(gpu_device):
    """The headline model itself (BASELINE metric: d24, C=1536, 24 heads, 24 blocks) + the full VQVAE, fp32 mode, token for token against
    the reference's recorded greedy trace (gen_d24_b2.npz: B=2, cfg 4, cond_type=[0,1]; control_var.py:356-565), image crops and means."""
    vae, m = build(24, F32, gpu_device)
    _gen_check(m, golden('gen_d24_b2'), 2, torch.tensor([3, 7]), 4.0, torch.tensor([0, 1]), 'd24 B=2 (headline model, full width)')


def test_d24_conditional_infer_fp32_matches_reference(gpu_device):
    """The headline model through conditional_infer_cfg (control_var.py:223-354): 4-branch CFG with cfg = (4, 4, 4) and the control tokens
    teacher-forced from the tokenised synthetic control images, token for token against the reference's trace (gen_d24_cmask.npz)."""
    vae, m = build(24, F32, gpu_device)
    g = golden('gen_d24_cmask')
    c_ids = split(t(g['c_ids']).long(), mf=1)
    _gen_check(m, g, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), torch.tensor([2, 3]), 'd24 conditional_infer_cfg (headline model)', four=True, c_mask=c_ids)


def test_d24_bf16_logits_along_the_reference_fp32_trace(gpu_device):
    """north_star's "within 1e-3 on bf16 logits" for the model the metric is quoted on, measured against the REFERENCE'S OWN bf16:
    gen_d24_bf16ref.npz holds the reference d24 walked under torch.autocast('cpu', bfloat16) along the greedy ids of its fp32 run (gen_d24_b2.npz) - its
    CFG-combined logits sit 1.12e-2 (max) / 2.1e-3 (RMS) of max|logit| from its fp32 ones and 70 of its 2 720 argmax ids move.  The HIP bf16 model is forced
    along the same ids; asserted: it is NO FARTHER from the reference's fp32 logits than the reference's autocast is (max and RMS over the same sampled
    logits, x 1.0), it flips no more greedy ids than the autocast does, and every flip happens at a reference margin below twice the autocast's own max
    distance (a bound that comes from the recording, not from this implementation)."""
    g, gb = golden('gen_d24_b2'), golden('gen_d24_bf16ref')
    vae, m = build(24, BF16, gpu_device)
    ids = split(t(g['ids']).long())
    m.autoregressive_infer_cfg(2, torch.tensor([3, 7]), g_seed=0, cfg=4.0, top_k=1, cond_type=torch.tensor([0, 1]), _force_idx=ids, _trace=True)
    tr = m.last_trace
    hip = torch.cat([x.float() for x in tr['logits']], dim=1)[:2, ::3, ::128].cpu()
    ref, rac = t(g['logit_samples']), t(gb['ref_autocast'])
    assert hip.shape == ref.shape == rac.shape, (hip.shape, ref.shape, rac.shape)
    assert float((t(gb['ref_fp32']) - ref).abs().max()) < 1e-4            # the autocast recording belongs to this fp32 trace
    amax = float(ref.abs().max())
    dmax, drms = [v / amax for v in _d(hip, ref)]
    amax_ac, arms_ac = [v / amax for v in _d(rac, ref)]
    print(f'[bf16] d24 along the reference fp32 trace: HIP max {dmax:.2e} / RMS {drms:.2e}; reference autocast {amax_ac:.2e} / {arms_ac:.2e} of max|logit| = {amax:.2f} (sampled logits)')
    record('gen_d24 bf16 vs reference fp32 (forced along the reference trace)', kind='bf16_logits', absmax=amax, hip_vs_ref_fp32=[dmax, drms],
           ref_autocast_vs_ref_fp32=[amax_ac, arms_ac], ref_autocast_flips=int(gb['flips_autocast_vs_fp32']))
    assert dmax <= amax_ac and drms <= arms_ac
    own = torch.cat(tr['idx'], dim=1).cpu()
    n, _ = check_ids(own, g['ids'], g['margin'], 2 * amax_ac * amax, 'gen_d24 bf16 greedy ids vs the reference fp32 trace', strict=False)
    assert n <= int(gb['flips_autocast_vs_fp32']), (n, int(gb['flips_autocast_vs_fp32']))


def test_forward_d24_bf16_among_the_references_bf16(gpu_device):
    """the d24 counterpart of test_forward_d12_bf16_among_the_references_bf16: forward_d24_bf16ref.npz holds the teacher-forced logits of the reference
    d24 (control_var.py:568-651) in fp32, under CPU bf16 autocast and from the oracle's bf16 emulation on one input (B = 1).  The HIP bf16 path must be no
    farther from the reference's fp32 logits than the reference's own autocast (max and RMS), and within 1e-3 RMS (north_star) of the emulation."""
    g = golden('forward_d24_bf16ref')
    vae, m = build(24, BF16, gpu_device)
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(1, 1358, 32, generator=gen).to(gpu_device)
    with torch.no_grad():
        hip = m(t(g['labels']), x, t(g['types']), True).float().cpu()[:, ::9, ::31]
    r = _fourway('forward_d24 bf16', hip, t(g['ref_fp32']), t(g['ref_autocast']), t(g['emu']), float(g['absmax']))
    assert r['hip_vs_ref_fp32'][0] <= r['ref_autocast_vs_ref_fp32'][0]
    assert r['hip_vs_ref_fp32'][1] <= r['ref_autocast_vs_ref_fp32'][1]
    assert r['hip_vs_emulation'][1] <= 1e-3


def test_d30_full_width_bf16_properties(gpu_device):
    """config 4 in the throughput mode: bit-reproducible; the KV-cached decode and the masked teacher-forced forward agree on every
    scale's logits (cfg = 0, forced ids); the four condition types give four different control maps for one label; conditional_infer_cfg
    keeps the forced control tokens."""
    vae, m = build(30, BF16, gpu_device)
    B = 4
    labels = torch.tensor([7, 7, 7, 7])
    a = m.autoregressive_infer_cfg(B, labels, g_seed=3, cfg=4.0, top_k=1, cond_type=None, _trace=True)
    ids = [x.clone() for x in m.last_trace['idx']]
    a2 = m.autoregressive_infer_cfg(B, labels, g_seed=3, cfg=4.0, top_k=1, cond_type=None, _trace=True)
    assert torch.equal(a, a2) and all(torch.equal(x, y) for x, y in zip(ids, m.last_trace['idx']))
    assert a.shape == (B, 3, 512, 256) and torch.isfinite(a).all() and float(a.min()) >= 0 and float(a.max()) <= 1
    flat = torch.cat(ids, dim=1)
    for i in range(B):
        for j in range(i + 1, B):
            assert (flat[i] != flat[j]).float().mean() > 0.05, (i, j)            # condition type changes the generation
    m.autoregressive_infer_cfg(B, labels, g_seed=0, cfg=0.0, top_k=1, cond_type=None, _force_idx=ids, _trace=True)
    inf_logits = torch.cat(m.last_trace['logits'], dim=1).float().cpu()
    h_c = vae.idxBl_to_h([i[:, :p * p] for i, p in zip(ids, PN)])
    h_i = vae.idxBl_to_h([i[:, p * p:] for i, p in zip(ids, PN)])
    x = torch.cat([torch.cat((u, v), dim=1) for u, v in zip(h_c, h_i)], dim=1)
    with torch.no_grad():
        fw = m(labels, x, torch.tensor([0, 1, 2, 3])).float().cpu()
    assert (fw - inf_logits).abs().max().item() < 3e-2 * fw.abs().max().item()
    assert (fw.argmax(-1) == inf_logits.argmax(-1)).float().mean().item() > 0.97
    ctrl = synth_images(B, 256, seed=4).to(gpu_device)
    c_ids = vae.img_to_idxBl(ctrl)
    b = m.conditional_infer_cfg(B, labels, g_seed=1, cfg=(4.0, 4.0, 4.0), top_k=900, top_p=0.96, cond_type=torch.tensor([0, 1, 2, 3]), c_mask=c_ids, _trace=True)
    assert b.shape == (B, 3, 512, 256) and torch.isfinite(b).all()
    want = vae.idxBl_to_img(c_ids, same_shape=True, last_one=True).add(1).mul(0.5).clamp(0, 1)
    assert (b[:, :, :256] - want).abs().max() < 2e-2                                # the control half IS the decoded c_mask


# ---------------------------------------------------------------------------------------------------------------- config 3
def _tokens(vae, dev, B, seed_img, seed_mask):
    images, masks = synth_images(B, 256, seed=seed_img).to(dev), synth_images(B, 256, seed=seed_mask).to(dev)
    mi = vae.img_to_idxBl(masks); mh = vae.idxBl_to_h(mi)
    ii = vae.img_to_idxBl(images); ih = vae.idxBl_to_h(ii)
    labels = torch.cat([torch.cat((a, b), 1) for a, b in zip(mi, ii)], dim=1)
    x = torch.cat([torch.cat((a, b), 1) for a, b in zip(mh, ih)], dim=1)
    return images, masks, x, labels


def test_d24_training_step_fp32_matches_reference(gpu_device):
    """one training step at d24 width, fp32 mode: loss, every parameter's gradient norm and a 64-element slice of every gradient against
    the reference's autograd (train_step_d24.npz; tiny VQVAE for the tokens, B=2, dropouts off as in the recording)"""
    g = golden('train_step_d24')
    vae, m = build(24, F32, gpu_device, ch=32)
    _, _, x, labels = _tokens(vae, gpu_device, 2, 6, 7)
    assert np.array_equal(labels.cpu().numpy(), g['labels'].astype(np.int64))
    eng = T.TrainEngine(m, drop_path=False)
    loss, _ = eng.forward_backward(torch.tensor([17, 403]), x, torch.tensor([2, 0]), labels)
    assert abs(loss.item() - float(g['loss'])) < 5e-5 * max(1.0, abs(float(g['loss'])))
    grads = eng.grads()
    names = [str(n) for n in g['names']]
    gn = t(g['gnorms'])
    worst, worst_name = 0.0, ''
    floor = 1e-3 * float(g['total_norm'])                # tensors whose whole gradient is below 0.1 % of the total are compared on that scale
    for i, n in enumerate(names):
        gg = grads[n]
        ref_n = gn[i].item()
        e = abs(gg.norm().item() - ref_n) / max(ref_n, floor)
        if e > worst:
            worst, worst_name = e, n
        ref_slice = t(g['g:' + n])
        got = gg.reshape(-1)[:: max(1, gg.numel() // 64)][:64].cpu()
        assert (got - ref_slice).abs().max() <= 2e-3 * max(ref_slice.abs().max().item(), 1e-5) + 1e-7, n
    print(f'd24 fp32 training step: worst relative gradient-norm error over {len(names)} parameters {worst:.2e} ({worst_name})')
    # measured 2.6e-3 (an adaLN generator weight: a rank-B outer product of sums over 1360 tokens with heavy cancellation); 2x that
    assert worst < 5e-3
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())).item()
    assert abs(total - float(g['total_norm'])) < 3e-3 * float(g['total_norm'])        # measured 1.3e-3 (dominated by the adaLN generator weights above)


def test_d24_training_step_bf16_properties(gpu_device):
    """config 3 shape in the throughput mode, B=4 per GPU: finite loss, bit-reproducible step (two identical models end up with identical
    parameters), bf16 gradients aligned with the fp32-mode gradients of the same batch (cosine per tensor), loss falls over three steps
    on a fixed batch, and the per-layer gradient slabs the data-parallel reducer would send are exactly the engine's gradients."""
    B = 4
    cls, types = torch.tensor([17, 403, 5, 999]), torch.tensor([2, 0, 1, 3])
    kw = dict(peak_lr=8e-5 * 256 / 512, weight_decay=0.08, sche='lin0', warmup_it=0, max_it=100, clip=2.0, drop_path=False)   # d24 yaml:17-24
    params = []
    for rep in range(2):
        vae, m = build(24, BF16, gpu_device, ch=32)
        images, masks, x, labels = _tokens(vae, gpu_device, B, 16, 17)
        tr = T.Trainer(m, vae, **kw)
        outs = [tr.step(images, masks, cls, types) for _ in range(3)]
        losses = [o['loss'].item() for o in outs]
        assert all(np.isfinite(losses)) and all(np.isfinite(o['grad_norm'].item()) for o in outs)
        params.append({k: v.clone() for k, v in m.state_dict().items()})
        if rep == 0:
            g_bf16 = {k: v.clone() for k, v in tr.engine.grads().items()}
            slabs = [b.clone() for b in tr.engine.buckets]
            x0, labels0 = x, labels
            first_losses = losses
    assert all(torch.equal(params[0][k], params[1][k]) for k in params[0]), 'training step is not bit-reproducible'
    assert first_losses[2] < first_losses[0], first_losses
    assert sum(s.numel() for s in slabs) >= sum(v.numel() for v in g_bf16.values())        # slabs cover every gradient (+ padding)
    # fp32-mode gradients of the SAME weights/batch as the third bf16 step started from: rebuild the state before it
    vae32, m32 = build(24, F32, gpu_device, ch=32)
    vae_b, m_b = build(24, BF16, gpu_device, ch=32)
    eng32, engb = T.TrainEngine(m32, drop_path=False), T.TrainEngine(m_b, drop_path=False)
    eng32.forward_backward(cls, x0, types, labels0)
    engb.forward_backward(cls, x0, types, labels0)
    g32, gb = eng32.grads(), engb.grads()
    worst = 1.0
    for n, p in m32.named_parameters():
        if p.numel() < 4096:
            continue
        cos = torch.nn.functional.cosine_similarity(g32[n].reshape(1, -1).double(), gb[n].reshape(1, -1).double()).item()
        worst = min(worst, cos)
    print(f'd24 training: worst per-tensor cosine(bf16 gradient, fp32 gradient) = {worst:.5f}')
    assert worst > 0.98
