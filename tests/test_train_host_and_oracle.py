"""CPU: training-step oracle against the reference fixture (A20), LR/WD schedule and parameter groups (A21)."""
import numpy as np
import pytest
import torch

from conftest import golden
from controlvar_amd import train as T
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VaeConfig, VarConfig, phi_index_map
from controlvar_amd.synth import synth_images, synth_vae_state, synth_var_state
from oracle import train_ref, vqvae_ref
from oracle.vqvae_ref import MSQuant

torch.set_num_threads(8)


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_lr_schedule_matches_reference_table():
    g = golden('lr_lin0')
    for row in g['table']:
        it, min_lr, max_lr, min_wd, max_wd = row
        lr, wd = T.lr_wd_factors('lin0', float(g['peak_lr']), float(g['wd']), float(g['wd_end']), int(it), int(g['wp_it']), int(g['max_it']),
                                 wp0=float(g['wp0']), wpe=float(g['wpe']))
        assert abs(lr - max_lr) < 1e-15 and abs(wd - max_wd) < 1e-15


def build_inputs(mask_first=True):
    sdv = synth_vae_state(VaeConfig(ch=32))
    msq = MSQuant(sdv, PN, phi_index_map(10))
    images, masks = synth_images(2, 256, seed=6), synth_images(2, 256, seed=7)
    with torch.no_grad():
        mi = vqvae_ref.img_to_idxBl(sdv, msq, masks); mh = msq.idx_to_var_input(mi)
        ii = vqvae_ref.img_to_idxBl(sdv, msq, images); ih = msq.idx_to_var_input(ii)
    if not mask_first:                                # train_control_var_hpu.py:192-195: image first
        mi, ii, mh, ih = ii, mi, ih, mh
    labels = torch.cat([torch.cat((a, b), 1) for a, b in zip(mi, ii)], dim=1)
    x = torch.cat([torch.cat((a, b), 1) for a, b in zip(mh, ih)], dim=1)
    return x, labels


TRAIN_CASES = {'d2': (VarConfig(depth=2), 0), 'd2v': (VarConfig(depth=2, shared_aln=True, type_pos=True), 5),
               'd2sa': (VarConfig(depth=2, sa_block=True, layer_scale=0.1), 7),
               'd2b': (VarConfig(depth=2, bidirectional=True, type_pos=True), 9)}         # image first (mask_first=False)


@pytest.mark.parametrize('tag', list(TRAIN_CASES))
def test_training_step_oracle_matches_reference(tag):
    """'d2v': the shared_aln + type_pos variant, 'd2sa': the SABlock variant"""
    g = golden(f'train_step_{tag}')
    cfg, wseed = TRAIN_CASES[tag]
    sd = synth_var_state(cfg, wseed)
    mask_first = tag != 'd2b'
    x, labels = build_inputs(mask_first)
    assert np.array_equal(labels.numpy(), g['labels'].astype(np.int64))
    assert (x[:, ::7] - t(g['x_sample'])).abs().max() < 2e-5
    loss, loss_tok, grads = train_ref.loss_and_grads(sd, cfg, torch.tensor([17, 403]), x, torch.tensor([2, 0]), labels, mask_first=mask_first)
    assert abs(loss.item() - float(g['loss'])) < 1e-5
    assert (loss_tok[::17] - t(g['loss_tok'])).abs().max() < 1e-4
    names = [str(n) for n in g['names']]
    for i, n in enumerate(names):
        gn = grads[n].norm().item()
        assert abs(gn - float(g['gnorms'][i])) < 2e-4 * max(1.0, float(g['gnorms'][i])), n
        sl = grads[n].reshape(-1)[:: max(1, grads[n].numel() // 64)][:64]
        assert (sl - t(g['g:' + n])).abs().max() < 2e-4 * max(1.0, float(np.abs(g['g:' + n]).max())), n
    total, coef = train_ref.clip_coef(grads, 2.0)
    assert abs(total - float(g['total_norm'])) < 1e-3 * float(g['total_norm'])
    # AdamW step with the reference's groups and schedule
    lr, wd = T.lr_wd_factors('lin0', 2e-3, 0.05, 0.01, 7, 20, 1000, wp0=0.005, wpe=0.01)
    assert abs(lr - g['lrs'][1]) < 1e-15 and abs(wd - g['lrs'][3]) < 1e-15
    nd = set(str(n) for n in g['nd_names'])
    for n in names:
        assert T.decays(n, sd[n].ndim) == (n not in nd), n
        p, _, _ = train_ref.adamw_update(sd[n], grads[n] * coef, torch.zeros_like(sd[n]), torch.zeros_like(sd[n]), 1, lr,
                                         wd if n not in nd else 0.0)
        sl = p.reshape(-1)[:: max(1, p.numel() // 64)][:64]
        assert (sl - t(g['p:' + n])).abs().max() < 1e-5, n


def test_filter_params_groups():
    from controlvar_amd import models
    vae = models.build_vae(ch=32)
    m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True)
    names, paras, groups = T.filter_params(m)
    assert len(names) == len(paras) == sum(len(g['params']) for g in groups)
    nd = {n for n, p in zip(names, paras) if not T.decays(n, p.ndim)}
    assert {'pos_1LC', 'pos_start', 'lvl_embed.weight', 'blocks.0.attn.q_bias', 'head.bias'} <= nd
    assert 'class_emb.weight' not in nd and 'blocks.1.ffn.fc1.weight' not in nd and 'cond_embed.weight' not in nd
