"""CPU: checkpoint wire format against the reference fixture tests/golden/checkpoint_d2.npz.

The fixture was produced by running the reference's own load_var_weight on a DDP-style VAR-d2 file built from the synth
recipe, and by dumping the layout of the reference optimizer's state_dict (tests/golden/make_golden.py::case_checkpoint).
"""
import hashlib
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import golden
from controlvar_amd import checkpoint as ckpt
from controlvar_amd import models
from controlvar_amd import train as T
from controlvar_amd.spec import VaeConfig, VarConfig
from controlvar_amd.synth import synth_vae_state, synth_var_state


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def make_models():
    vae = models.build_vae(ch=32)
    cvar = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, cond_drop_rate=0.0)
    cvar.load_state_dict(synth_var_state(VarConfig(depth=2), 0), strict=True)
    var = models.build_var(vae, depth=2)
    var.load_state_dict(synth_var_state(VarConfig(depth=2, mask_factor=1, control=False), 3), strict=True)
    return vae, var, cvar


@pytest.mark.parametrize('interpos', [False, True])
def test_var_checkpoint_surgery_matches_reference(tmp_path, interpos):
    g = golden('checkpoint_d2')
    vae, var, cvar = make_models()
    path = os.path.join(tmp_path, 'var_d2.pth')
    torch.save({'model_state_dict': OrderedDict(('module.' + k, v) for k, v in var.state_dict().items())}, path)
    res = ckpt.load_var_weight(cvar, path, interpos=interpos)
    assert sorted(res.missing_keys) == sorted(['lvl_1L', 'pos_start', 'attn_bias_for_masking', 'cond_embed.weight'])
    assert list(res.unexpected_keys) == []
    tag = 'ip1' if interpos else 'ip0'
    sd = cvar.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f'{tag}_keys']]
    for k, want in zip(sd.keys(), g[f'{tag}_sha']):
        assert sha(sd[k]) == str(want), k
    np.testing.assert_array_equal(sd['pos_1LC'][0, ::37, ::5].numpy(), g[f'{tag}_pos'])


def test_surgery_rejects_a_non_var_file():
    _, var, cvar = make_models()
    sd = dict(var.state_dict()); del sd['lvl_1L']
    with pytest.raises(KeyError):
        ckpt.load_var_weight(cvar, sd)


def test_read_state_forms_and_strict_load(tmp_path):
    vae, var, _ = make_models()
    plain = var.state_dict()
    wrapped = {'model_state_dict': OrderedDict(('module.' + k, v) for k, v in plain.items()), 'epoch': 3}
    for form in (plain, wrapped):
        got = ckpt.read_state(form)
        assert list(got.keys()) == list(plain.keys())
    var2 = models.build_var(vae, depth=2)
    var2.load_state_dict(synth_var_state(VarConfig(depth=2, mask_factor=1, control=False), 9), strict=True)
    ckpt.load_weights(var2, wrapped)
    for k, v in var2.state_dict().items():
        assert torch.equal(v, plain[k]), k
    bad = dict(plain); bad.pop('head.bias')
    with pytest.raises(RuntimeError):
        ckpt.load_weights(var2, bad)


def test_vqvae_loads_a_file_saved_with_another_scale_count():
    """vqvae.py:106-109: ema_vocab_hit_SV of a different scale count is replaced by the model's own"""
    vae = models.build_vae(ch=32)
    sd = synth_vae_state(VaeConfig(ch=32), 0)
    sd = dict(sd)
    sd['quantize.ema_vocab_hit_SV'] = torch.ones(7, 4096)
    ckpt.load_weights(vae, sd)
    assert vae.quantize.ema_vocab_hit_SV.shape[0] == 10


def test_optimizer_state_layout_matches_reference_adamw():
    g = golden('checkpoint_d2')
    _, _, cvar = make_models()
    opt = T.FusedAdamW(cvar, lr=2e-3, weight_decay=0.05)
    assert opt.state_dict()['state'] == {}                                   # torch: no state before the first step
    opt.steps = 2
    osd = opt.state_dict()
    order = [n for grp in opt.param_groups for n in grp['names']]
    assert order == [str(n) for n in g['opt_order']]
    assert [len(grp['params']) for grp in osd['param_groups']] == list(g['opt_group_sizes'])
    assert [grp['wd_sc'] for grp in osd['param_groups']] == list(g['opt_wd_sc'])
    assert sorted(osd['param_groups'][0].keys()) == [str(k) for k in g['opt_group_keys']]
    assert sorted(osd['state'][0].keys()) == [str(k) for k in g['opt_state_keys']]
    assert str(osd['state'][0]['step'].dtype) == str(g['opt_step_dtype']) and float(osd['state'][0]['step']) == float(g['opt_step'])
    assert [i for grp in osd['param_groups'] for i in grp['params']] == list(range(len(order)))


def test_optimizer_state_round_trips_through_torch_adamw():
    _, _, cvar = make_models()
    names, paras, groups = T.filter_params(cvar)
    topt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    gen = torch.Generator().manual_seed(0)
    for _ in range(3):
        for p in paras:
            p.grad = torch.randn(p.shape, generator=gen) * 1e-3
        topt.step()
    fused = T.FusedAdamW(cvar, lr=5e-4, weight_decay=0.01)
    fused.load_state_dict(topt.state_dict())                                 # reference -> this package
    assert fused.steps == 3
    assert [g['lr'] for g in fused.param_groups] == [1e-3, 1e-3]
    assert [g['weight_decay'] for g in fused.param_groups] == [g['weight_decay'] for g in topt.param_groups]
    by_id = {id(p): n for n, p in cvar.named_parameters()}
    for grp in topt.param_groups:
        for p in grp['params']:
            m, v = fused.state[by_id[id(p)]]
            assert torch.equal(m, topt.state[p]['exp_avg']) and torch.equal(v, topt.state[p]['exp_avg_sq'])
    topt2 = torch.optim.AdamW(T.filter_params(cvar)[2], lr=7e-4, betas=(0.9, 0.95), weight_decay=0.0)
    topt2.load_state_dict(fused.state_dict())                                # this package -> reference
    for grp, grp2 in zip(topt.param_groups, topt2.param_groups):
        assert grp2['lr'] == grp['lr'] and grp2['weight_decay'] == grp['weight_decay'] and grp2['wd_sc'] == grp['wd_sc']
        for p in grp['params']:
            assert torch.equal(topt2.state[p]['exp_avg'], topt.state[p]['exp_avg'])
            assert float(topt2.state[p]['step']) == 3.0


def test_optimizer_load_rejects_mismatched_groups():
    _, var, cvar = make_models()
    a, b = T.FusedAdamW(cvar, lr=1e-3), T.FusedAdamW(var, lr=1e-3)
    a.steps = b.steps = 1
    with pytest.raises(ValueError):
        a.load_state_dict(b.state_dict())


def test_save_checkpoint_and_resume(tmp_path):
    vae, _, cvar = make_models()
    opt = T.FusedAdamW(cvar, lr=1e-3, weight_decay=0.05)
    opt.steps = 5
    for m, v in opt.state.values():
        m.fill_(0.25); v.fill_(0.5)
    p1 = ckpt.save_checkpoint(cvar, opt, epoch=2, step=500, save_dir=str(tmp_path))
    p2 = ckpt.save_checkpoint(cvar, opt, epoch=2, step=500, save_dir=str(tmp_path), latest=True)
    assert os.path.basename(p1) == 'checkpoint_step_500.pth' and os.path.basename(p2) == 'checkpoint_step_latest.pth'
    raw = torch.load(p1, map_location='cpu')
    assert sorted(raw.keys()) == ['epoch', 'model_state_dict', 'optimizer_state_dict', 'step']
    fresh = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True)
    fresh.load_state_dict(synth_var_state(VarConfig(depth=2), 11), strict=True)
    opt2 = T.FusedAdamW(fresh, lr=3e-4)
    assert ckpt.resume(fresh, opt2, p1) == (500, 3)                          # epoch + 1 for a step file
    assert ckpt.resume(fresh, opt2, p2) == (500, 2)                          # 'latest' resumes inside the epoch
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, cvar.state_dict()[k]), k
    assert opt2.steps == 5 and all(float(m.mean()) == 0.25 and float(v.mean()) == 0.5 for m, v in opt2.state.values())
    raw['model_state_dict'].pop('head.bias')
    with pytest.raises(RuntimeError):
        ckpt.resume(fresh, opt2, raw)


def test_separator_surgery_layout():
    """load_var_weight with separator (train_control_var_hpu.py:504-517,523-533): deterministic parts of the result - control-half position
    rows copied, image-half rows the constant 1 (upstream's `x * -1 if mpos else 1` precedence), head grown to V + 18 rows with the
    pretrained rows first and zero bias behind - and the state loads into a separator model with only the rebuilt tensors missing"""
    import torch
    from controlvar_amd import checkpoint as ckpt, models
    from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN
    vae = models.build_vae(ch=32)
    var = models.build_var(vae, depth=2)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, separator=True)
    sd = ckpt.var_to_control_var_state(ckpt.read_state({'model_state_dict': {'module.' + k: v for k, v in var.state_dict().items()}}), PN,
                                       separator=True, vocab_size=4096)
    pos, src = sd['pos_1LC'][0], var.state_dict()['pos_1LC'][0]
    assert pos.shape == (1378, 128) and sd['head.weight'].shape == (4114, 128) and sd['head.bias'].shape == (4114,)
    at, o = 0, 0
    for i, pn in enumerate(PN):
        sp = 1 if i else 0
        n = pn * pn
        assert torch.equal(pos[o:o + n], src[at:at + n])
        assert torch.equal(pos[o + n + sp:o + 2 * n + sp], torch.ones(n, 128))
        at += n
        o += 2 * (n + sp)
    assert torch.equal(sd['head.weight'][:4096], var.state_dict()['head.weight']) and sd['head.bias'][4096:].abs().max() == 0
    assert 0 < sd['head.weight'][4096:].abs().max() < 0.02 * 6 * (1 / 128 / 3) ** 0.5          # trunc_normal(std) * 0.02 (bounds +-2 are absolute)
    res = ckpt.load_var_weight(m, {'model_state_dict': var.state_dict()})
    assert set(res.missing_keys) == {'lvl_1L', 'pos_start', 'attn_bias_for_masking', 'cond_embed.weight', 'special_embed.weight'}
    assert m.state_dict()['pos_1LC'].shape == (1, 1378, 128)
