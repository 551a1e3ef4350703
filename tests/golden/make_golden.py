#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference's own ``models`` package (read-only, no bytecode written), loads the
deterministic synthetic weights of controlvar_amd.synth into the REFERENCE modules
(strict=True, which also pins controlvar_amd.spec's key/shape tables), runs the reference
and records its outputs as small fixtures under tests/golden/*.npz.  Only inputs' seeds and
the reference's outputs are stored - no reference source, no weights.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

The reference has no tests or golden vectors of its own, so these
fixtures are what pins oracle/ (tests/test_oracle_*.py) and, through it, the HIP path.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

import models as ref_models                                    # the reference
import models.control_var as ref_cv
import models.var as ref_v
from models import VQVAE, build_control_var, build_var
from models.control_var import ControlVAR

from controlvar_amd.spec import DEFAULT_PATCH_NUMS, VaeConfig, VarConfig
from controlvar_amd.synth import synth_images, synth_vae_state, synth_var_state

torch.set_num_threads(8)
PN = DEFAULT_PATCH_NUMS


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f'  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


def make_vae(ch, seed=0):
    vae = quiet(VQVAE, vocab_size=4096, z_channels=32, ch=ch, test_mode=True, share_quant_resi=4, v_patch_nums=PN)
    vae.load_state_dict(synth_vae_state(VaeConfig(ch=ch), seed), strict=True)
    return vae.eval()


def make_cvar(vae, cfg: VarConfig, seed=0):
    if cfg.control:
        if cfg.embed_dim:   # explicit small width (used for the depth==30 cos-attn case)
            m = quiet(ControlVAR, vae_local=vae, patch_nums=PN, depth=cfg.depth, embed_dim=cfg.C, num_heads=cfg.H,
                      flash_if_available=False, fused_if_available=False, mask_factor=cfg.mask_factor,
                      bidirectional=False, separate_decoding=False, separator=False, type_pos=False, indep=False,
                      multi_cond=cfg.multi_cond, cond_drop_rate=0.0)
        else:
            m = quiet(build_control_var, vae, depth=cfg.depth, patch_nums=PN, mask_type='interleave_append' if cfg.mask_factor == 2 else 'replace',
                      cond_drop_rate=0.0, multi_cond=cfg.multi_cond, flash_if_available=False, fused_if_available=False,
                      shared_aln=cfg.shared_aln, type_pos=cfg.type_pos, aln=-1 if cfg.sa_block else 1, layer_scale=cfg.layer_scale,
                      bidirectional=cfg.bidirectional, separate_decoding=cfg.separate_decoding, indep=cfg.indep, separator=cfg.separator)
    else:
        m = quiet(build_var, vae, depth=cfg.depth, patch_nums=PN, flash_if_available=False, fused_if_available=False, shared_aln=cfg.shared_aln)
        m.cond_drop_rate = 0.0
    m.load_state_dict(synth_var_state(cfg, seed), strict=True)
    if getattr(cfg, 'separator', False):
        # upstream bug: special_embed (18 rows) is indexed with V + k (control_var.py:549,606) -> IndexError in forward() and inference.
        # The fixtures are recorded with the evidently intended index k: this wraps the INSTANCE's embedding call, the reference source
        # runs unmodified otherwise.
        emb, V = m.special_embed, m.V
        orig = emb.forward
        emb.forward = lambda idx: orig(torch.where(idx >= V, idx - V, idx))
    return m.eval()


class CaptureIdx:
    """Record the ids returned by sample_with_top_k_top_p_ at every stage (the ids BEFORE
    conditional_infer_cfg's teacher-forcing overwrite) and the CFG-combined logits."""

    def __init__(self, module):
        self.module, self.orig = module, module.sample_with_top_k_top_p_
        self.idx, self.logit_samples, self.margins = [], [], []

    def __enter__(self):
        def wrapped(logits, *a, **k):
            lg = logits.detach().clone()
            t2 = lg.topk(2, dim=-1).values
            self.margins.append((t2[..., 0] - t2[..., 1]).clone())
            self.logit_samples.append(lg[:2, :, ::128].clone())
            out = self.orig(logits, *a, **k)
            self.idx.append(out[:, :, 0].clone())
            return out
        self.module.sample_with_top_k_top_p_ = wrapped
        return self

    def __exit__(self, *exc):
        self.module.sample_with_top_k_top_p_ = self.orig


def img_stats(img):
    return dict(mean=img.mean(dim=(2, 3)), std=img.std(dim=(2, 3)), crop=img[:, :, 100:116, 60:76].clone(),
                crop2=img[:, :, -20:-4, 200:216].clone())


# ------------------------------------------------------------------------------- cases
def case_interp():
    """F.interpolate area / bicubic on random maps for every scale (pins oracle/interp.py)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    out = {}
    f = torch.randn(2, 3, 16, 16, generator=g)
    out['f'] = f
    for p in PN[:-1]:
        out[f'area_{p}'] = F.interpolate(f, size=(p, p), mode='area')
        h = torch.randn(2, 3, p, p, generator=g)
        out[f'h_{p}'] = h
        out[f'bicubic_{p}'] = F.interpolate(h, size=(16, 16), mode='bicubic')
    save('interp', **out)


def case_tokenizer(ch, nimg, tag):
    vae = make_vae(ch)
    img = synth_images(nimg, 256, seed=1)
    t0 = time.time()
    with torch.no_grad():
        f = vae.quant_conv(vae.encoder(img))
        ids = vae.img_to_idxBl(img, v_patch_nums=PN)
        fhats = vae.quantize.f_to_idxBl_or_fhat(f, to_fhat=True, v_patch_nums=PN)
        var_in = torch.cat(vae.idxBl_to_h(ids), dim=1)
        rec = vae.idxBl_to_img(ids, same_shape=True, last_one=True)
        rec2 = vae.fhat_to_img(fhats[-1].clone())
    print(f'  [{tag}] reference encode+decode {time.time() - t0:.1f}s;  rec==rec2 maxdiff {(rec - rec2).abs().max():.2e}')
    st = img_stats(rec)
    save(f'tokenizer_{tag}', nimg=nimg, ch=ch, f=f, ids=torch.cat(ids, dim=1).to(torch.int16), fhat_last=fhats[-1],
         fhat_s3=fhats[3], var_in=var_in[:, ::5].contiguous(), rec_mean=st['mean'], rec_std=st['std'], rec_crop=st['crop'], rec_crop2=st['crop2'])


def case_next_input():
    """get_next_autoregressive_input for every si on random (f_hat, h) (quant.py:243-260)."""
    vae = make_vae(32)
    g = torch.Generator().manual_seed(5)
    out = {}
    for si, pn in enumerate(PN):
        f_hat = torch.randn(1, 32, 16, 16, generator=g)
        h = torch.randn(1, 32, pn, pn, generator=g)
        out[f'fhat_in_{si}'] = f_hat.clone()
        out[f'h_{si}'] = h
        with torch.no_grad():
            f2, nxt = vae.quantize.get_next_autoregressive_input(si, len(PN), f_hat, h)
        out[f'fhat_out_{si}'] = f2.clone()
        out[f'next_{si}'] = nxt.clone()
    save('next_input', **out)


def case_block(cos):
    """One AdaLNSABlock (C=128, H=2) with the KV cache over two stages, and masked training
    mode on 10 tokens (basic_var.py:203-210)."""
    from functools import partial
    from models.basic_var import AdaLNSABlock
    import torch.nn as nn
    C, H = 128, 2
    cfg = VarConfig(depth=30 if cos else 2, embed_dim=C, num_heads=H)
    sd = synth_var_state(cfg, seed=3)
    blk = quiet(AdaLNSABlock, block_idx=0, last_drop_p=0, embed_dim=C, cond_dim=C, shared_aln=False,
                norm_layer=partial(nn.LayerNorm, eps=1e-6), num_heads=H, tau=4, cos_attn=cos,
                flash_if_available=False, fused_if_available=False).eval()
    blk.load_state_dict({k[len('blocks.0.'):]: v for k, v in sd.items() if k.startswith('blocks.0.')}, strict=True)
    g = torch.Generator().manual_seed(9)
    cond = torch.randn(3, C, generator=g)
    x0 = torch.randn(3, 2, C, generator=g)
    x1 = torch.randn(3, 8, C, generator=g)
    xm = torch.randn(3, 10, C, generator=g)
    lvl = torch.tensor([0, 0, 1, 1, 1, 1, 1, 1, 1, 1]).view(1, 10, 1)
    bias = torch.where(lvl >= lvl.transpose(1, 2), 0., -torch.inf).reshape(1, 1, 10, 10)
    with torch.no_grad():
        blk.attn.kv_caching(True)
        y0 = blk(x0, cond, None)
        y1 = blk(x1, cond, None)
        blk.attn.kv_caching(False)
        ym = blk(xm, cond, bias)
    save('block_cos' if cos else 'block', cond=cond, x0=x0, x1=x1, xm=xm, y0=y0, y1=y1, ym=ym)


def case_forward(depth, tag, mf=2):
    """Teacher-forced logits of a small model (control_var.py:568-651 / var.py:209-253)."""
    vae = make_vae(32)
    cfg = VarConfig(depth=depth, mask_factor=mf, control=(mf == 2), multi_cond=(mf == 2))
    m = make_cvar(vae, cfg)
    B = 2
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
    labels = torch.tensor([3, 977])
    types = torch.tensor([1, 2])
    with torch.no_grad():
        logits = m(labels, x, types, True) if mf == 2 else m(labels, x)
    t2 = logits.topk(2, dim=-1).values
    save(f'forward_{tag}', depth=depth, mf=mf, labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(),
         argmax=logits.argmax(-1).to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float())


def _run_generate(m, module, B, labels, cfg_scale, cond_type=None, four=False, c_mask=None, c_img=None, top_k=1, top_p=0.0, seed=0, more_smooth=False):
    with CaptureIdx(module) as cap, torch.no_grad():
        if four:
            img = m.conditional_infer_cfg(B=B, label_B=labels, g_seed=seed, cfg=cfg_scale, top_k=top_k, top_p=top_p,
                                          cond_type=cond_type, c_mask=c_mask, c_img=c_img, more_smooth=more_smooth)
        elif module is ref_cv:
            img = m.autoregressive_infer_cfg(B=B, label_B=labels, g_seed=seed, cfg=cfg_scale, top_k=top_k, top_p=top_p, cond_type=cond_type,
                                             more_smooth=more_smooth)
        else:
            img = m.autoregressive_infer_cfg(B=B, label_B=labels, g_seed=seed, cfg=cfg_scale, top_k=top_k, top_p=top_p)
    ids = torch.cat(cap.idx, dim=1)
    st = img_stats(img)
    return dict(ids=ids.to(torch.int16), margin=torch.cat(cap.margins, dim=1), logit_samples=torch.cat(cap.logit_samples, dim=1)[:, ::3].contiguous(),
                img_mean=st['mean'], img_std=st['std'], img_crop=st['crop'], img_crop2=st['crop2'])


def case_generate_tiny():
    """Greedy decode traces of depth-2 models over the tiny VQVAE (ch=32)."""
    vae = make_vae(32)
    cfg = VarConfig(depth=2)
    m = make_cvar(vae, cfg)
    # (a) B=2 explicit cond types
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
    save('gen_d2_b2', **r)
    # (b) B=4, cond_type=None -> [0,1,2,3]
    r = _run_generate(m, ref_cv, 4, torch.tensor([1, 10, 100, 999]), 4.0, cond_type=None)
    save('gen_d2_b4none', **r)
    # (c) conditional_infer_cfg with c_mask / c_img from synthetic control images
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids)
    save('gen_d2_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (3.0, 2.0, 1.0), cond_type=torch.tensor([2, 3]), four=True, c_img=c_ids)
    save('gen_d2_cimg', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)
    # (d) stochastic defaults of the reference (top_k=900, top_p=0.96), CPU generator
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=900, top_p=0.96, seed=42)
    save('gen_d2_b2_sampled', **r)
    # (e) plain VAR, mask_factor 1
    cfgv = VarConfig(depth=2, mask_factor=1, control=False, multi_cond=False)
    mv = make_cvar(vae, cfgv)
    r = _run_generate(mv, ref_v, 2, torch.tensor([3, 7]), 4.0)
    save('gen_var_d2_b2', **r)
    # (f) cos-attn (depth == 30 forces it), narrow width
    cfgc = VarConfig(depth=30, embed_dim=128, num_heads=2)
    mc = make_cvar(vae, cfgc)
    r = _run_generate(mc, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([3, 0]))
    save('gen_d30n_b2', **r)


def case_generate_d12():
    """BASELINE config 1: d12 ControlVAR + full VQVAE (ch=160), B=2, greedy, cfg=4."""
    vae = make_vae(160)
    m = make_cvar(vae, VarConfig(depth=12))
    t0 = time.time()
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
    print(f'  d12 B=2 reference generate {time.time() - t0:.1f}s')
    save('gen_d12_b2', **r)


def case_sampler():
    """-inf pattern of the top-k/top-p filter for fixed logits (helpers.py:8-15)."""
    from models.helpers import sample_with_top_k_top_p_
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(2, 6, 4096, generator=g) * 3
    out = {}
    for (k, p) in [(900, 0.96), (0, 0.5), (50, 0.0), (1, 0.0)]:
        lg = logits.clone()
        rng = torch.Generator().manual_seed(1)
        idx = sample_with_top_k_top_p_(lg, top_k=k, top_p=p, rng=rng)[:, :, 0]
        out[f'kept_{k}_{p}'] = np.packbits(torch.isfinite(lg).numpy(), axis=-1)
        out[f'idx_{k}_{p}'] = idx
    save('sampler', **out)


def case_lr():
    """lr_wd_annealing 'lin0' table (utils/lr_control.py:10-64)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_lr_control', '/root/reference/utils/lr_control.py')   # utils/__init__ needs wandb
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lr_wd_annealing = mod.lr_wd_annealing
    import torch.nn as nn
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{'params': [p], 'wd_sc': 1.0, 'lr_sc': 1.0}], lr=1.0)
    rows = []
    max_it, wp_it = 1000, 10
    for it in [0, 1, 5, 9, 10, 11, 50, 59, 60, 61, 100, 500, 900, 999]:
        lo = lr_wd_annealing('lin0', opt, 4e-5, 0.08, 0.08, it, wp_it, max_it, wp0=0.005, wpe=0.01)
        rows.append([it] + [float(v) for v in lo])
    save('lr_lin0', table=np.array(rows, dtype=np.float64), peak_lr=4e-5, wd=0.08, wd_end=0.08, wp_it=wp_it, max_it=max_it, wp0=0.005, wpe=0.01)


def case_bidirectional():
    """bidirectional=True (+ type_pos so that type_1L_ is exercised).  forward(mask_first=False) and one
    autoregressive_infer_cfg whose python-`random` draw (random.seed(2) -> 0.956 >= 0.5) selects the image-first order."""
    import random
    vae = make_vae(32)
    cfg = VarConfig(depth=2, bidirectional=True, type_pos=True)
    m = make_cvar(vae, cfg, seed=9)
    g = torch.Generator().manual_seed(24)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
    labels, types = torch.tensor([11, 640]), torch.tensor([1, 2])
    with torch.no_grad():
        logits = m(labels, x, types, False)
    t2 = logits.topk(2, dim=-1).values
    save('forward_d2b', keys=np.array(list(m.state_dict().keys())), labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(),
         argmax=logits.argmax(-1).to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float())
    random.seed(2)
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
    save('gen_d2b_b2', **r)


def case_train_step(cfg=None, tag='d2', wseed=0, mask_first=True):
    """A20: one training step of the reference on a depth-2 ControlVAR + tiny VQVAE (train_control_var_hpu.py:157-250):
    tokenise control + image, interleave (mask first), teacher-forced forward, CE mean, backward, clip 2.0, AdamW with
    filter_params groups and lr_wd_annealing('lin0').  Dropouts off (cond_drop_rate=0, eval-mode DropPath)."""
    import importlib.util
    from itertools import chain
    spec = importlib.util.spec_from_file_location('ref_lr_control', '/root/reference/utils/lr_control.py')
    lrc = importlib.util.module_from_spec(spec); spec.loader.exec_module(lrc)
    vae = make_vae(32)
    cfg = cfg or VarConfig(depth=2)
    m = make_cvar(vae, cfg, seed=wseed)  # eval(): DropPath is identity; cond_drop_rate = 0
    for p_ in m.parameters():
        p_.requires_grad_(True)
    images, masks = synth_images(2, 256, seed=6), synth_images(2, 256, seed=7)
    cls, types = torch.tensor([17, 403]), torch.tensor([2, 0])
    with torch.no_grad():
        mask_ids = vae.img_to_idxBl(masks, v_patch_nums=PN); mask_h = vae.idxBl_to_h(mask_ids)
        img_ids = vae.img_to_idxBl(images, v_patch_nums=PN); img_h = vae.idxBl_to_h(img_ids)
    if mask_first:
        labels_list = list(chain.from_iterable(zip(mask_ids, img_ids)))
        h_list = list(chain.from_iterable(zip(mask_h, img_h)))
    else:                                             # train_control_var_hpu.py:192-195 (bidirectional, image first)
        labels_list = list(chain.from_iterable(zip(img_ids, mask_ids)))
        h_list = list(chain.from_iterable(zip(img_h, mask_h)))
    x = torch.cat(h_list, dim=1)
    if getattr(cfg, 'separator', False):             # train_control_var_hpu.py:214-224: a separator label behind every half of scales >= 1
        mapping = [i for i in range(18)] if mask_first else [i + 1 if i % 2 == 0 else i - 1 for i in range(18)]
        new = [labels_list[0], labels_list[1]]
        for i, label in enumerate(labels_list[2:]):
            new.extend([label, label.new_ones(label.shape[0], 1) * (mapping[i] + 4096)])
        labels_list = new
    labels = torch.cat(labels_list, dim=1)
    logits = m(cls, x, types, mask_first)
    loss_tok = torch.nn.CrossEntropyLoss(reduction='none')(logits.view(-1, logits.size(-1)), labels.view(-1))
    loss = loss_tok.mean()
    loss.backward()
    names = [n for n, _ in m.named_parameters()]
    gnorms = torch.stack([p_.grad.norm() for _, p_ in m.named_parameters()])
    sl = {}
    for n, p_ in m.named_parameters():
        g = p_.grad
        sl['g:' + n] = g.reshape(-1)[:: max(1, g.numel() // 64)][:64].clone()
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 2.0)
    _, paras, groups = lrc.filter_params(m, nowd_keys={'cls_token', 'start_token', 'task_token', 'cfg_uncond', 'pos_embed', 'pos_1LC',
                                                        'pos_start', 'start_pos', 'lvl_embed', 'gamma', 'beta', 'ada_gss', 'moe_bias', 'scale_mul'})
    nd_names = [n for n, p_ in m.named_parameters() if any(p_ is q for q in groups[[g['wd_sc'] for g in groups].index(0.)]['params'])]
    opt = torch.optim.AdamW(groups, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05)
    lrs = lrc.lr_wd_annealing('lin0', opt, 2e-3, 0.05, 0.01, 7, 20, 1000, wp0=0.005, wpe=0.01)
    opt.step()
    after = {}
    for n, p_ in m.named_parameters():
        after['p:' + n] = p_.detach().reshape(-1)[:: max(1, p_.numel() // 64)][:64].clone()
    save(f'train_step_{tag}', loss=loss.detach(), loss_tok=loss_tok.detach()[::17].clone(), labels=labels.to(torch.int16), x_sample=x[:, ::7].clone(),
         names=np.array(names), nd_names=np.array(nd_names), gnorms=gnorms, total_norm=total_norm, lrs=np.array(lrs, dtype=np.float64), **sl, **after)


def case_checkpoint():
    """N1: the reference's own load_var_weight (train_control_var_hpu.py:472-534; the function body is lifted out of the
    script by ast HERE, at generation time only, because the script itself imports Habana modules) applied to a DDP-style
    VAR-d2 checkpoint -> per-key SHA-256 of the ControlVAR state that results; plus the layout of the reference
    optimizer's state_dict (group order / index lists / keys) for the wire-format test."""
    import argparse, ast, hashlib, importlib.util, math, tempfile
    from collections import OrderedDict
    import torch.nn as nn
    src = open('/root/reference/train_control_var_hpu.py').read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'load_var_weight'][0]
    ns = dict(torch=torch, nn=nn, math=math, OrderedDict=OrderedDict)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'load_var_weight', 'exec'), ns)
    vae = make_vae(32)
    var_plain = make_cvar(vae, VarConfig(depth=2, mask_factor=1, control=False), seed=3)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'var_d2.pth')
        torch.save({'model_state_dict': OrderedDict(('module.' + k, v) for k, v in var_plain.state_dict().items())}, path)
        for interpos in (False, True):
            m = make_cvar(vae, VarConfig(depth=2), seed=0)
            args = argparse.Namespace(var_pretrained_path=path, embed_dim=128, mask_type='interleave_append', interpos=interpos,
                                      separator=False, mpos=False, v_patch_nums=PN, vocab_size=4096)
            quiet(ns['load_var_weight'], m, args)
            tag = 'ip1' if interpos else 'ip0'
            keys = list(m.state_dict().keys())
            out[f'{tag}_keys'] = np.array(keys)
            out[f'{tag}_sha'] = np.array([hashlib.sha256(v.contiguous().numpy().tobytes()).hexdigest() for v in m.state_dict().values()])
            out[f'{tag}_pos'] = m.state_dict()['pos_1LC'][0, ::37, ::5].clone()
    spec = importlib.util.spec_from_file_location('ref_lr_control', '/root/reference/utils/lr_control.py')
    lrc = importlib.util.module_from_spec(spec); spec.loader.exec_module(lrc)
    m = make_cvar(vae, VarConfig(depth=2), seed=0)
    for p_ in m.parameters():
        p_.requires_grad_(True)
    names, paras, groups = lrc.filter_params(m, nowd_keys={'cls_token', 'start_token', 'task_token', 'cfg_uncond', 'pos_embed', 'pos_1LC',
                                                            'pos_start', 'start_pos', 'lvl_embed', 'gamma', 'beta', 'ada_gss', 'moe_bias', 'scale_mul'})
    opt = torch.optim.AdamW(groups, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05)
    for p_ in paras:
        p_.grad = torch.full_like(p_, 1e-3)
    opt.step(); opt.step()
    osd = opt.state_dict()
    by_id = {id(p_): n for n, p_ in m.named_parameters()}
    order = [by_id[id(p_)] for g in groups for p_ in g['params']]
    save('checkpoint_d2', opt_order=np.array(order), opt_group_sizes=np.array([len(g['params']) for g in osd['param_groups']]),
         opt_group_keys=np.array(sorted(osd['param_groups'][0].keys())), opt_wd_sc=np.array([g['wd_sc'] for g in osd['param_groups']]),
         opt_state_keys=np.array(sorted(osd['state'][0].keys())), opt_step=osd['state'][0]['step'].clone(),
         opt_step_dtype=np.array(str(osd['state'][0]['step'].dtype)), **out)


from controlvar_amd.synth import PREPROC_CASES, synth_photo_pair as preproc_inputs      # noqa: E402


def case_preprocess():
    """N2: the third-party pieces of the reference's input pipeline, recorded from the libraries themselves (Pillow here;
    torch for the nearest-neighbour ignore mask): Image.resize(LANCZOS) to the torchvision F.resize(288) size, the default
    (BICUBIC) cond.resize(image.size), and F.interpolate(mode='nearest') of the background mask (imagenetC.py:147-178)."""
    import hashlib
    import PIL
    from PIL import Image
    import torch.nn.functional as F
    from controlvar_amd.preprocess import resized_size
    out = {'pillow_version': np.array(PIL.__version__)}
    for h, w, seed in PREPROC_CASES:
        img, cond = preproc_inputs(h, w, seed)
        nh, nw = resized_size(h, w, 288)
        big = np.asarray(Image.fromarray(img).resize((nw, nh), Image.LANCZOS)) if (nh, nw) != (h, w) else img
        c1 = np.asarray(Image.fromarray(cond).resize((w, h)))                      # PIL default filter
        c2 = np.asarray(Image.fromarray(c1).resize((nw, nh), Image.LANCZOS)) if (nh, nw) != (h, w) else c1
        tag = f'{h}x{w}'
        for name, arr in (('img288', big), ('cond_fit', c1), ('cond288', c2)):
            out[f'{tag}_{name}_sha'] = np.array(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest())
            out[f'{tag}_{name}_shape'] = np.array(arr.shape)
            out[f'{tag}_{name}_sample'] = arr[::37, ::29].copy()
    # ignore masks and annotation painting by the reference's OWN TEXT: datasets/imagenetC.py cannot be imported (torchvision,
    # pycocotools, tqdm are absent), so - as case_checkpoint does for load_var_weight - the function definitions (:15-37) and the
    # ``if cond_type == 'mask': ... else: ...`` statement of __getitem__ (:152-181) are lifted out of the file by ast at generation
    # time and compiled unmodified.  The only stand-in is ``mask_utils.decode`` (pycocotools, third party): the synthetic annotations
    # carry their DECODED masks as ``segmentation``, so decode is the identity on arrays.
    import ast, types
    src = open('/root/reference/datasets/imagenetC.py').read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('process_anns', 'create_color_map')]
    assert [f.name for f in fns] == ['process_anns', 'create_color_map'] and fns[0].lineno == 15 and fns[1].lineno == 31
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'ImagenetCDataset'][0]
    getitem = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '__getitem__'][0]
    blocks = [n for n in getitem.body if isinstance(n, ast.If) and any(isinstance(x, ast.Name) and x.id == 'ignore_mask' for x in ast.walk(n))]
    assert len(blocks) == 1 and blocks[0].lineno == 152, [b.lineno for b in blocks]
    wrapper = ast.parse('def ignore_block(self, cond, cond_type):\n    pass\n    return ignore_masks, ignore_masks_').body[0]
    wrapper.body[0] = blocks[0]
    mod = ast.fix_missing_locations(ast.Module(body=fns + [wrapper], type_ignores=[]))
    ns = dict(np=np, torch=torch, F=F, mask_utils=types.SimpleNamespace(decode=lambda seg: np.asarray(seg, dtype=np.uint8)))
    exec(compile(mod, 'imagenetC.py(ast)', 'exec'), ns)
    from controlvar_amd.synth import synth_annotations
    colormap = ns['create_color_map']()
    out['colormap'] = colormap
    for seed in (2, 3, 4):
        anns = [{'area': a['area'], 'segmentation': a['_mask']} for a in synth_annotations(seed, n=8)]
        canvas = ns['process_anns'](anns, 512, colormap)
        assert canvas.max() > 0 and canvas.dtype == np.float64
        out[f'anns{seed}_canvas'] = canvas.astype(np.uint8)                       # imagenetC.py:144 casts the same way
        out[f'anns{seed}_kept'] = np.array(sum(a['area'] >= 5000 for a in anns))
    _, cond = preproc_inputs(375, 500, 0)
    c = np.asarray(Image.fromarray(cond).resize((256, 256), Image.NEAREST)).astype(np.float32)
    ct = ((torch.from_numpy(c).permute(2, 0, 1) / 255.0) - 0.5) / 0.5
    res = {}
    for sep in (False, True):
        me = types.SimpleNamespace(v_patch_nums=PN, separator=sep)
        a, b = ns['ignore_block'](me, ct, 'mask')
        a2, b2 = ns['ignore_block'](me, ct, 'depth')
        assert float(a2.min()) == 1.0 and a2.shape == a.shape == b.shape == b2.shape == ((1378,) if sep else (1360,))
        res['ignore_mask' + ('_sep' if sep else '')] = a
        res['ignore_mask_' + ('_sep' if sep else '')] = b
    save('preprocess', ign_cond=ct, **res, **out)


def case_sa_block():
    """aln < 0 -> SABlock (affine LayerNorms, layer-scale gammas) + Sequential(LN, Linear) head; depth 2."""
    vae = make_vae(32)
    for tag, cfg, seed in (('d2sa', VarConfig(depth=2, sa_block=True, layer_scale=0.1), 7), ('d2sa0', VarConfig(depth=2, sa_block=True), 8)):
        m = make_cvar(vae, cfg, seed=seed)
        g = torch.Generator().manual_seed(23)
        x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
        labels, types = torch.tensor([9, 500]), torch.tensor([3, 1])
        with torch.no_grad():
            logits = m(labels, x, types, True)
        t2 = logits.topk(2, dim=-1).values
        save(f'forward_{tag}', keys=np.array(list(m.state_dict().keys())), labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(),
             argmax=logits.argmax(-1).to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float())
        if tag == 'd2sa':
            r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
            save('gen_d2sa_b2', **r)


def case_variants():
    """shared_aln (SharedAdaLin + per-block ada_gss) and type_pos (type_embed), depth 2, tiny VQVAE.
    Records the reference's state_dict key order for both models, teacher-forced logits and greedy decode traces - note
    that upstream adds the type embedding in forward() (all rows) and autoregressive_infer_cfg (scales >= 1) but NOT in
    conditional_infer_cfg; the fixtures pin exactly that."""
    vae = make_vae(32)
    cfg = VarConfig(depth=2, shared_aln=True, type_pos=True)
    m = make_cvar(vae, cfg, seed=5)
    keys = list(m.state_dict().keys())
    B = 2
    g = torch.Generator().manual_seed(22)
    x = torch.randn(B, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
    labels, types = torch.tensor([4, 321]), torch.tensor([2, 0])
    with torch.no_grad():
        logits = m(labels, x, types, True)
    t2 = logits.topk(2, dim=-1).values
    save('forward_d2v', keys=np.array(keys), labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(),
         argmax=logits.argmax(-1).to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float())
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
    save('gen_d2v_b2', **r)
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 3.0, 2.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids)
    save('gen_d2v_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)
    cfgv = VarConfig(depth=2, mask_factor=1, control=False, multi_cond=False, shared_aln=True)
    mv = make_cvar(vae, cfgv, seed=6)
    r = _run_generate(mv, ref_v, 2, torch.tensor([3, 7]), 4.0)
    save('gen_var_d2s_b2', keys=np.array(list(mv.state_dict().keys())), **r)


def case_forward_d12():
    """BASELINE config 2 anchor: the reference's teacher-forced logits at d12 width (C=768, 12 heads, 12 blocks), B=2."""
    vae = make_vae(32)
    cfg = VarConfig(depth=12)
    m = make_cvar(vae, cfg)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
    labels, types = torch.tensor([12, 800]), torch.tensor([3, 1])
    with torch.no_grad():
        logits = m(labels, x, types, True)
    t2 = logits.topk(2, dim=-1).values
    save('forward_d12', labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(), argmax=logits.argmax(-1).to(torch.int16),
         margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float(), absmax=logits.abs().amax())


def case_generate_d12_bf16emu():
    """BASELINE config 2: d12, bf16 weights / activations with fp32 accumulation and fp32 logits.  The reference cannot run this
    mode on a CPU (bf16 autocast hits addmm_ dtype errors), so the trace is the ORACLE's, with its bf16 storage
    points (oracle.vqvae_ref.Prec(True)) - the oracle itself is pinned to the reference in fp32 by every other fixture, and its fp32
    d12 trace equals gen_d12_b2.npz.  B=8 rows: labels arange(8), types arange(8) % 4, greedy, cfg 4."""
    from oracle import var_ref
    from oracle.vqvae_ref import MSQuant, Prec
    from controlvar_amd.spec import phi_index_map
    cfg = VarConfig(depth=12)
    sdv, sd = synth_vae_state(VaeConfig(ch=160)), synth_var_state(cfg)
    msq = MSQuant(sdv, PN, phi_index_map(10))
    B = 8
    labels, types = torch.arange(B) % 1000, torch.arange(B) % 4
    trace = {}
    t0 = time.time()
    with torch.no_grad():
        var_ref.generate(sd, cfg, msq, B, labels, 4.0, top_k=1, cond_type=types, prec=Prec(True), trace=trace)
    print(f'  oracle d12 B={B} bf16-emulated generate {time.time() - t0:.1f}s')
    ids = torch.cat(trace['idx'], dim=1)
    lg = torch.cat(trace['logits'], dim=1)                              # (B, L, V) CFG-combined
    t2 = lg.topk(2, dim=-1).values
    save('gen_d12_bf16emu', ids=ids.to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), logit_samples=lg[:4, :, 5::128].contiguous(),
         absmax_per_scale=torch.stack([t.abs().amax() for t in trace['logits']]), labels=labels, types=types)


class ReplayIdx(CaptureIdx):
    """CaptureIdx that makes the reference CONTINUE from recorded ids: the sampler's result is replaced by `forced[stage]`, the logits of
    every stage are still recorded.  Lets the fp32 reference be walked along the trace its bf16-autocast run produced."""

    def __init__(self, module, forced):
        super().__init__(module)
        self.forced = forced

    def __enter__(self):
        def wrapped(logits, *a, **k):
            lg = logits.detach().clone().float()
            t2 = lg.topk(2, dim=-1).values
            self.margins.append((t2[..., 0] - t2[..., 1]).clone())
            self.logit_samples.append(lg.clone())
            out = self.forced[len(self.idx)].clone().unsqueeze(-1)
            self.idx.append(out[:, :, 0].clone())
            return out
        self.module.sample_with_top_k_top_p_ = wrapped
        return self


def _dist(a, b):
    d = (a.double() - b.double())
    return [float(d.abs().max()), float(d.pow(2).mean().sqrt())]


def case_forward_d12_bf16ref():
    """north_star: "within 1e-3 on bf16 logits".  The reference's OWN bf16 behaviour, recorded: teacher-forced logits of the d12-width
    model (inputs of forward_d12.npz) from the reference in fp32, from the reference under torch.autocast('cpu', torch.bfloat16)
    (control_var.py:568-651 runs there; only the tokenizer ENCODE fails under CPU autocast, quant.py:205-206, and it is not on this
    call), and from the oracle with bf16 storage points.  Stored: the [::9, ::31] sample of each and the full-tensor max / RMS distances
    between them, so that the GPU test can place the HIP bf16 path among them."""
    from oracle import var_ref
    from oracle.vqvae_ref import Prec
    vae = make_vae(32)
    cfg = VarConfig(depth=12)
    m = make_cvar(vae, cfg)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=g)
    labels, types = torch.tensor([12, 800]), torch.tensor([3, 1])
    with torch.no_grad():
        l32 = m(labels, x, types, True).float()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            lac = m(labels, x, types, True).float()
        lemu = var_ref.forward_logits(synth_var_state(cfg), cfg, labels, x, types, prec=Prec(True))
    S = (slice(None), slice(None, None, 9), slice(None, None, 31))
    t2 = lac.topk(2, dim=-1).values
    print(f'  max|logit| {float(l32.abs().max()):.3f}; autocast-fp32 {_dist(lac, l32)}, emu-fp32 {_dist(lemu, l32)}, emu-autocast {_dist(lemu, lac)}')
    save('forward_d12_bf16ref', labels=labels, types=types, absmax=l32.abs().amax(), ref_fp32=l32[S].contiguous(), ref_autocast=lac[S].contiguous(),
         emu=lemu[S].contiguous(), d_autocast_fp32=_dist(lac, l32), d_emu_fp32=_dist(lemu, l32), d_emu_autocast=_dist(lemu, lac),
         argmax_autocast=lac.argmax(-1).to(torch.int16), margin_autocast=(t2[..., 0] - t2[..., 1]),
         argmax_agree_autocast_fp32=float((lac.argmax(-1) == l32.argmax(-1)).float().mean()),
         argmax_agree_emu_fp32=float((lemu.argmax(-1) == l32.argmax(-1)).float().mean()))


def case_generate_d12_bf16ref():
    """BASELINE config 2 pinned to the reference's bf16: d12 + full VQVAE, B=8 (labels arange(8), types arange(8) % 4), greedy, cfg 4,
    autoregressive_infer_cfg (control_var.py:356-565) under torch.autocast('cpu', torch.bfloat16): per-stage ids and CFG-combined logits.
    Then the fp32 reference and the bf16-emulating oracle are walked along THOSE ids (sampler replaced by the recorded ids), so the three
    logit sets belong to the same token path.  Stored per stage: ids, margins and the [rows 0..3, :, 5::128] logit sample of all three, and
    the full-tensor max / RMS distances per scale."""
    from oracle import var_ref
    from oracle.vqvae_ref import MSQuant, Prec
    from controlvar_amd.spec import phi_index_map
    vae = make_vae(160)
    cfg = VarConfig(depth=12)
    m = make_cvar(vae, cfg)
    B = 8
    labels, types = torch.arange(B) % 1000, torch.arange(B) % 4
    kw = dict(B=B, label_B=labels, g_seed=0, cfg=4.0, top_k=1, top_p=0.0, cond_type=types)
    t0 = time.time()
    full = {}

    class Cap(CaptureIdx):
        def __enter__(s2):
            def wrapped(logits, *a, **k):
                lg = logits.detach().clone().float()
                t2 = lg.topk(2, dim=-1).values
                s2.margins.append((t2[..., 0] - t2[..., 1]).clone())
                s2.logit_samples.append(lg.clone())
                out = s2.orig(logits, *a, **k)
                s2.idx.append(out[:, :, 0].clone())
                return out
            s2.module.sample_with_top_k_top_p_ = wrapped
            return s2

    with Cap(ref_cv) as cap, torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        img = m.autoregressive_infer_cfg(**kw)
    print(f'  reference d12 B={B} autocast(bf16) generate {time.time() - t0:.1f}s')
    ids = [i.clone() for i in cap.idx]
    t0 = time.time()
    with ReplayIdx(ref_cv, ids) as rep, torch.no_grad():
        m.autoregressive_infer_cfg(**kw)
    print(f'  reference fp32 along the same ids {time.time() - t0:.1f}s')
    sdv, sd = synth_vae_state(VaeConfig(ch=160)), synth_var_state(cfg)
    trace = {}
    with torch.no_grad():
        var_ref.generate(sd, cfg, MSQuant(sdv, PN, phi_index_map(10)), B, labels, 4.0, top_k=1, cond_type=types, prec=Prec(True), trace=trace, force_idx=ids)
    lac, l32, lemu = cap.logit_samples, rep.logit_samples, trace['logits']
    d_ac32 = [_dist(a, b) for a, b in zip(lac, l32)]
    d_emu32 = [_dist(a, b) for a, b in zip(lemu, l32)]
    d_emuac = [_dist(a, b) for a, b in zip(lemu, lac)]
    amax = torch.stack([t.abs().amax() for t in l32])
    for si in range(len(PN)):
        print(f'   scale {si}: max|logit| {float(amax[si]):.2f}  autocast-fp32 {d_ac32[si][0]:.3e}/{d_ac32[si][1]:.3e}  emu-fp32 {d_emu32[si][0]:.3e}/{d_emu32[si][1]:.3e}  emu-autocast {d_emuac[si][0]:.3e}/{d_emuac[si][1]:.3e}')
    samp = lambda L: torch.cat([t[:4, :, 5::128] for t in L], dim=1).contiguous()
    st = img_stats(img.float())
    save('gen_d12_bf16ref', ids=torch.cat(ids, dim=1).to(torch.int16), margin_autocast=torch.cat(cap.margins, dim=1), margin_fp32=torch.cat(rep.margins, dim=1),
         ref_autocast=samp(lac), ref_fp32=samp(l32), emu=samp(lemu), absmax_per_scale=amax, d_autocast_fp32=d_ac32, d_emu_fp32=d_emu32, d_emu_autocast=d_emuac,
         labels=labels, types=types, img_mean=st['mean'], img_std=st['std'], img_crop=st['crop'])


def case_generate_d30():
    """BASELINE config 4: d30 (cos-attention, C=1920, 30 heads) at FULL width with the full VQVAE, B=4, cond_type=None -> the four
    condition types [0,1,2,3] (control_var.py:387-389), cfg 4, greedy; and conditional_infer_cfg with cfg=(4,4,4) (the script
    default, train_control_var_hpu.py:77) and c_mask from synthetic control images."""
    vae = make_vae(160)
    m = make_cvar(vae, VarConfig(depth=30))
    t0 = time.time()
    r = _run_generate(m, ref_cv, 4, torch.tensor([1, 10, 100, 999]), 4.0, cond_type=None)
    print(f'  d30 B=4 reference generate {time.time() - t0:.1f}s')
    save('gen_d30_b4none', **r)
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    t0 = time.time()
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids)
    print(f'  d30 B=2 reference conditional_infer_cfg {time.time() - t0:.1f}s')
    save('gen_d30_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)


def case_generate_d24():
    """The headline model itself (BASELINE metric: 256^2 autoregressive_infer_cfg, d24): d24 ControlVAR (C=1536, 24 heads) at FULL
    width + full VQVAE (ch=160), B=2, greedy, cfg 4, cond_type=[0,1] recorded from the reference (control_var.py:356-565)."""
    vae = make_vae(160)
    m = make_cvar(vae, VarConfig(depth=24))
    t0 = time.time()
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
    print(f'  d24 B=2 reference generate {time.time() - t0:.1f}s')
    save('gen_d24_b2', **r)
    # the same model through conditional_infer_cfg (control_var.py:223-354): 4-branch CFG with cfg = (4, 4, 4) - the script default,
    # train_control_var_hpu.py:77 - and the control tokens teacher-forced from synthetic control images
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    t0 = time.time()
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids)
    print(f'  d24 B=2 reference conditional_infer_cfg {time.time() - t0:.1f}s')
    save('gen_d24_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)


def case_d24_bf16ref():
    """the bf16 yardstick for the model the metric is quoted on.  (a) gen_d24_bf16ref: the reference d24 + full VQVAE
    walked under torch.autocast('cpu', bfloat16) ALONG the greedy ids of its own fp32 run (gen_d24_b2.npz, committed; control_var.py:356-565 with the
    sampler's result replaced by the recorded ids): per-stage CFG-combined logits, sampled exactly like gen_d24_b2's `logit_samples`
    ([rows 0..1, every third position, vocabulary ::128]), the full-tensor max / RMS distance to the fp32 logits per scale, and the autocast run's own
    argmax ids + margins.  (b) forward_d24_bf16ref: teacher-forced logits (control_var.py:568-651) of the same model in fp32, under autocast and from the
    oracle's bf16 emulation on one input, as forward_d12_bf16ref."""
    from oracle import var_ref
    from oracle.vqvae_ref import Prec
    g = np.load(os.path.join(HERE, 'gen_d24_b2.npz'))
    vae = make_vae(160)
    cfg = VarConfig(depth=24)
    m = make_cvar(vae, cfg)
    ids_all = torch.from_numpy(g['ids'].astype(np.int64))
    ids, o = [], 0
    for p_ in PN:
        ids.append(ids_all[:, o:o + 2 * p_ * p_].clone()); o += 2 * p_ * p_
    kw = dict(B=2, label_B=torch.tensor([3, 7]), g_seed=0, cfg=4.0, top_k=1, top_p=0.0, cond_type=torch.tensor([0, 1]))
    t0 = time.time()
    with ReplayIdx(ref_cv, ids) as r32, torch.no_grad():
        m.autoregressive_infer_cfg(**kw)
    print(f'  reference d24 fp32 along its own trace {time.time() - t0:.1f}s')
    t0 = time.time()
    with ReplayIdx(ref_cv, ids) as rac, torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        m.autoregressive_infer_cfg(**kw)
    print(f'  reference d24 autocast(bf16) along the fp32 trace {time.time() - t0:.1f}s')
    l32, lac = r32.logit_samples, rac.logit_samples
    samp = lambda L: torch.cat([t_[:2, :, ::128] for t_ in L], dim=1)[:, ::3].contiguous()
    assert float((samp(l32) - torch.from_numpy(g['logit_samples'])).abs().max()) < 1e-4          # the replay reproduces the committed fp32 recording
    d = [_dist(a, b) for a, b in zip(lac, l32)]
    amax = torch.stack([t_.abs().amax() for t_ in l32])
    arg_ac = torch.cat([t_.argmax(-1) for t_ in lac], dim=1)
    print('  autocast vs fp32 per scale (max / RMS, relative to max|logit|):', ' '.join(f'{d[i][0] / float(amax[i]):.2e}/{d[i][1] / float(amax[i]):.2e}' for i in range(len(PN))))
    print(f'  overall: max {max(x[0] for x in d) / float(amax.max()):.3e}; autocast argmax != fp32 ids: {int((arg_ac != ids_all).sum())} of {ids_all.numel()}')
    save('gen_d24_bf16ref', ref_autocast=samp(lac), ref_fp32=samp(l32), absmax_per_scale=amax, d_autocast_fp32=d, absmax=amax.max(),
         d_autocast_fp32_sampled=_dist(samp(lac), samp(l32)), argmax_autocast=arg_ac.to(torch.int16), margin_autocast=torch.cat(rac.margins, dim=1),
         flips_autocast_vs_fp32=np.array(int((arg_ac != ids_all).sum())))
    # (b) teacher-forced forward, B = 1
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(1, cfg.pyramid.L - cfg.pyramid.first_l, 32, generator=gen)
    labels, types = torch.tensor([77]), torch.tensor([2])
    t0 = time.time()
    with torch.no_grad():
        f32 = m(labels, x, types, True).float()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            fac = m(labels, x, types, True).float()
        femu = var_ref.forward_logits(synth_var_state(cfg), cfg, labels, x, types, prec=Prec(True))
    print(f'  d24 forward fp32 / autocast / emulation {time.time() - t0:.1f}s; max|logit| {float(f32.abs().max()):.3f}; autocast-fp32 {_dist(fac, f32)}, emu-fp32 {_dist(femu, f32)}, emu-autocast {_dist(femu, fac)}')
    S = (slice(None), slice(None, None, 9), slice(None, None, 31))
    save('forward_d24_bf16ref', labels=labels, types=types, absmax=f32.abs().amax(), ref_fp32=f32[S].contiguous(), ref_autocast=fac[S].contiguous(),
         emu=femu[S].contiguous(), d_autocast_fp32=_dist(fac, f32), d_emu_fp32=_dist(femu, f32), d_emu_autocast=_dist(femu, fac))


def case_train_step_d24():
    """BASELINE config 3 anchor: one reference training step at d24 width (C=1536, 24 blocks), B=2, tiny VQVAE for the tokens."""
    case_train_step(VarConfig(depth=24), 'd24', 0)


def case_tokenizer_alt():
    """vqvae.py:73-75 with a caller-chosen scale list: ids of the tiny VQVAE for v_patch_nums (1,2,4,8,16) and (1,3,5,7,9,11,13,16)"""
    vae = make_vae(32)
    img = synth_images(2, 256, seed=1)
    out = {}
    with torch.no_grad():
        for tag, pns in (('a', (1, 2, 4, 8, 16)), ('b', (1, 3, 5, 7, 9, 11, 13, 16))):
            ids = vae.img_to_idxBl(img, v_patch_nums=pns)
            rec = vae.img_to_recon(img, v_patch_nums=pns, last_one=True)
            out[f'pns_{tag}'] = np.array(pns)
            out[f'ids_{tag}'] = torch.cat(ids, dim=1).to(torch.int16)
            out[f'rec_crop_{tag}'] = rec[:, :, 100:116, 60:76].clone()
            recs = vae.img_to_recon(img, v_patch_nums=pns, last_one=False)          # one reconstruction per scale of the chosen list
            assert len(recs) == len(pns) and torch.equal(recs[-1], rec)
            out[f'recs_crop_{tag}'] = torch.stack([r[:, :, 100:116, 60:76] for r in recs])
            out[f'recs_mean_{tag}'] = torch.stack([r.mean(dim=(2, 3)) for r in recs])
    save('tokenizer_alt', **out)


def case_lowres():
    """vqvae.py:91-104 / quant.py:156-182 with all_to_max_scale / same_shape = False: the reference's own per-scale f_hat list (each scale at
    its own resolution) and the images decoded from it (tiny VQVAE, B=2), plus embed_to_fhat(all_to_max_scale=True) for the same embeddings."""
    vae = make_vae(32)
    img = synth_images(2, 256, seed=5)
    with torch.no_grad():
        ids = vae.img_to_idxBl(img, v_patch_nums=PN)
        ms_h = [vae.quantize.embedding(i).transpose(1, 2).view(2, 32, pn, pn) for i, pn in zip(ids, PN)]
        low = vae.quantize.embed_to_fhat(ms_h, all_to_max_scale=False, last_one=False)
        full = vae.quantize.embed_to_fhat(ms_h, all_to_max_scale=True, last_one=False)
        imgs = vae.idxBl_to_img(ids, same_shape=False, last_one=False)
        last = vae.idxBl_to_img(ids, same_shape=False, last_one=True)
    out = dict(ids=torch.cat(ids, dim=1).to(torch.int16))
    for si, pn in enumerate(PN):
        assert low[si].shape == (2, 32, pn, pn) and imgs[si].shape == (2, 3, 16 * pn, 16 * pn)
        out[f'low_{si}'] = low[si].clone()
        out[f'img_mean_{si}'] = imgs[si].mean(dim=(2, 3))
        out[f'img_crop_{si}'] = imgs[si][:, :, :16, :16].clone()               # the top-left 16x16 of every scale's image
        out[f'full_mean_{si}'] = full[si].mean(dim=(2, 3))
    out['full_last'] = full[-1].clone()
    assert torch.equal(last, imgs[-1])
    save('lowres', **out)


def _forward_fixture(m, cfg, tag, xseed, labels, types, mask_first=True):
    g = torch.Generator().manual_seed(xseed)
    x = torch.randn(2, len(cfg.pyramid.code_positions()) - cfg.pyramid.first_l, 32, generator=g)       # code tokens only (no separators)
    with torch.no_grad():
        logits = m(labels, x, types, mask_first)
    t2 = logits.topk(2, dim=-1).values
    save(f'forward_{tag}', keys=np.array(list(m.state_dict().keys())), labels=labels, types=types, logits_sample=logits[:, ::9, ::31].contiguous(),
         argmax=logits.argmax(-1).to(torch.int16), margin=(t2[..., 0] - t2[..., 1]), lsum=logits.double().sum(-1).float())


def case_separate_decoding():
    """separate_decoding (per scale the control half is decoded before the image half) without and with indep:
    masked teacher-forced logits, the two-pass inference branch (control_var.py:428-485) and the indep branch, which applies
    the training mask's rows at inference (:497); conditional_infer_cfg for the indep model (:283)."""
    vae = make_vae(32)
    for tag, cfg, seed in (('d2s', VarConfig(depth=2, separate_decoding=True), 11), ('d2si', VarConfig(depth=2, separate_decoding=True, indep=True), 12)):
        m = make_cvar(vae, cfg, seed=seed)
        _forward_fixture(m, cfg, tag, 25, torch.tensor([8, 450]), torch.tensor([1, 3]))
        r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
        save(f'gen_{tag}_b2', **r)
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 3.0, 2.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids)
    save('gen_d2si_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)


def case_more_smooth():
    """more_smooth=True - Gumbel-softmax soft code embeddings instead of E[idx] (control_var.py:511-515,
    helpers.py:22-36), drawn from the model's CPU generator after the id draw: joint branch with the sampling defaults, greedy
    (top_k=1 masks all but the maximum in place, so the soft embedding collapses to E[argmax]), the 4-branch conditional form and the
    two-pass branch."""
    vae = make_vae(32)
    m = make_cvar(vae, VarConfig(depth=2))
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=900, top_p=0.96, seed=42, more_smooth=True)
    save('gen_d2_smooth', **r)
    r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=1, seed=1, more_smooth=True)
    save('gen_d2_smooth_greedy', **r)
    ctrl = synth_images(2, 256, seed=4)
    with torch.no_grad():
        c_ids = vae.img_to_idxBl(ctrl, v_patch_nums=PN)
    r = _run_generate(m, ref_cv, 2, torch.tensor([5, 6]), (4.0, 4.0, 4.0), cond_type=torch.tensor([2, 3]), four=True, c_mask=c_ids, top_k=900, top_p=0.96,
                      seed=7, more_smooth=True)
    save('gen_d2_smooth_cmask', c_ids=torch.cat(c_ids, dim=1).to(torch.int16), **r)
    ms = make_cvar(vae, VarConfig(depth=2, separate_decoding=True), seed=11)
    r = _run_generate(ms, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]), top_k=900, top_p=0.96, seed=42, more_smooth=True)
    save('gen_d2s_smooth', **r)


def case_separator():
    """separator=True (18 special tokens, head V + 18, special_embed), recorded with the index shim of make_cvar:
    teacher-forced logits and the joint inference branch, alone and combined with separate_decoding + indep (the two combinations whose
    inference runs upstream once the index is fixed; the two-pass branch still fails there with a shape error)."""
    vae = make_vae(32)
    for tag, cfg, seed in (('d2p', VarConfig(depth=2, separator=True), 13), ('d2psi', VarConfig(depth=2, separator=True, separate_decoding=True, indep=True), 14)):
        m = make_cvar(vae, cfg, seed=seed)
        _forward_fixture(m, cfg, tag, 26, torch.tensor([9, 451]), torch.tensor([0, 2]))
        r = _run_generate(m, ref_cv, 2, torch.tensor([3, 7]), 4.0, cond_type=torch.tensor([0, 1]))
        save(f'gen_{tag}_b2', **r)


CASES = {
    'interp': case_interp,
    'tok_tiny': lambda: case_tokenizer(32, 3, 'ch32'),
    'tok_full': lambda: case_tokenizer(160, 2, 'ch160'),
    'next_input': case_next_input,
    'block': lambda: case_block(False),
    'block_cos': lambda: case_block(True),
    'fwd_d2': lambda: case_forward(2, 'd2'),
    'fwd_var_d2': lambda: case_forward(2, 'var_d2', mf=1),
    'gen_tiny': case_generate_tiny,
    'gen_d12': case_generate_d12,
    'sampler': case_sampler,
    'lr': case_lr,
    'train': case_train_step,
    'checkpoint': case_checkpoint,
    'preprocess': case_preprocess,
    'variants': case_variants,
    'bidirectional': case_bidirectional,
    'train_bidirectional': lambda: case_train_step(VarConfig(depth=2, bidirectional=True, type_pos=True), 'd2b', 9, mask_first=False),
    'sa_block': case_sa_block,
    'train_sa_block': lambda: case_train_step(VarConfig(depth=2, sa_block=True, layer_scale=0.1), 'd2sa', 7),
    'train_variants': lambda: case_train_step(VarConfig(depth=2, shared_aln=True, type_pos=True), 'd2v', 5),
    'tok_alt': case_tokenizer_alt,
    'lowres': case_lowres,
    'separator': case_separator,
    'train_separator': lambda: case_train_step(VarConfig(depth=2, separator=True), 'd2p', 13),
    'separate_decoding': case_separate_decoding,
    'more_smooth': case_more_smooth,
    'train_separate_decoding': lambda: case_train_step(VarConfig(depth=2, separate_decoding=True, indep=True), 'd2si', 12),
    'fwd_d12': case_forward_d12,
    'gen_d12_bf16emu': case_generate_d12_bf16emu,
    'fwd_d12_bf16ref': case_forward_d12_bf16ref,
    'gen_d12_bf16ref': case_generate_d12_bf16ref,
    'gen_d30': case_generate_d30,
    'gen_d24': case_generate_d24,
    'train_d24': case_train_step_d24,
    'd24_bf16ref': case_d24_bf16ref,
}

if __name__ == '__main__':
    names = sys.argv[1:] or list(CASES)
    for n in names:
        print(f'[{n}]')
        t0 = time.time()
        CASES[n]()
        print(f'  done in {time.time() - t0:.1f}s')
