"""A5 backward / A20 on MI355X: loss and every parameter gradient of one training step against the CPU oracle
(torch autograd over the restated forward) and against the fixture recorded from the reference itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden  # noqa: E402
from controlvar_amd import models  # noqa: E402
from controlvar_amd import train as T  # noqa: E402
from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VaeConfig, VarConfig, phi_index_map  # noqa: E402
from controlvar_amd.synth import synth_images, synth_vae_state, synth_var_state  # noqa: E402
from oracle import train_ref  # noqa: E402


def t(a):
    return torch.from_numpy(np.asarray(a))


def make(cfg, dtype, dev):
    vae = models.build_vae(ch=32, compute_dtype=dtype).to(dev)
    if cfg.control:
        m = models.ControlVAR(vae, depth=cfg.depth, embed_dim=cfg.C, num_heads=cfg.H, mask_factor=2, multi_cond=True, patch_nums=PN,
                              compute_dtype=dtype, cond_drop_rate=0.0)
    else:
        m = models.VAR(vae, depth=cfg.depth, embed_dim=cfg.C, num_heads=cfg.H, patch_nums=PN, compute_dtype=dtype, cond_drop_rate=0.0)
    return vae, m.to(dev)


def tokenize(vae, dev):
    images, masks = synth_images(2, 256, seed=6).to(dev), synth_images(2, 256, seed=7).to(dev)
    mi = vae.img_to_idxBl(masks); mh = vae.idxBl_to_h(mi)
    ii = vae.img_to_idxBl(images); ih = vae.idxBl_to_h(ii)
    labels = torch.cat([torch.cat((a, b), 1) for a, b in zip(mi, ii)], dim=1)
    x = torch.cat([torch.cat((a, b), 1) for a, b in zip(mh, ih)], dim=1)
    return x, labels


def test_training_step_fp32_matches_reference_fixture(gpu_device):
    """tokenise (HIP) -> interleave -> forward -> CE -> backward, fp32 mode, against train_step_d2.npz (the reference)."""
    g = golden('train_step_d2')
    cfg = VarConfig(depth=2)
    vae, m = make(cfg, torch.float32, gpu_device)
    x, labels = tokenize(vae, gpu_device)
    assert np.array_equal(labels.cpu().numpy(), g['labels'].astype(np.int64))
    eng = T.TrainEngine(m, drop_path=False)
    loss, loss_tok = eng.forward_backward(torch.tensor([17, 403]), x, torch.tensor([2, 0]), labels)
    assert abs(loss.item() - float(g['loss'])) < 2e-5
    assert (loss_tok.cpu()[::17] - t(g['loss_tok'])).abs().max() < 2e-4
    grads = eng.grads()
    names = [str(n) for n in g['names']]
    assert set(names) == set(grads)
    worst = 0.0
    for i, n in enumerate(names):
        gr = grads[n].cpu()
        ref_norm = float(g['gnorms'][i])
        assert abs(gr.norm().item() - ref_norm) < 1e-3 * max(1.0, ref_norm), (n, gr.norm().item(), ref_norm)
        sl = gr.reshape(-1)[:: max(1, gr.numel() // 64)][:64]
        ref = t(g['g:' + n])
        err = (sl - ref).abs().max().item() / max(1.0, float(ref.abs().max()))
        worst = max(worst, err)
        assert err < 1e-3, (n, err)
    print(f'worst relative gradient-slice error vs the reference: {worst:.2e}')


@pytest.mark.parametrize('kind', ['control', 'var'])
def test_all_gradients_against_oracle_fp32(gpu_device, kind):
    """full tensors of every gradient vs autograd over the oracle (d2; ControlVAR and plain VAR), with an ignore mask"""
    cfg = VarConfig(depth=2) if kind == 'control' else VarConfig(depth=2, mask_factor=1, control=False, multi_cond=False)
    vae, m = make(cfg, torch.float32, gpu_device)
    sd = synth_var_state(cfg)
    B, L, fl = 2, cfg.pyramid.L, cfg.pyramid.first_l
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, L - fl, 32, generator=gen)
    tg = torch.randint(0, 4096, (B, L), generator=gen)
    im = (torch.rand(B, L, generator=gen) > 0.3).float()
    cls, ty = torch.tensor([5, 999]), torch.tensor([1, 3])
    loss_r, _, grads_r = train_ref.loss_and_grads(sd, cfg, cls, x, ty if kind == 'control' else None, tg, im)
    eng = T.TrainEngine(m, drop_path=False)
    loss, _ = eng.forward_backward(cls, x.to(gpu_device), ty, tg.to(gpu_device), im.to(gpu_device))
    assert abs(loss.item() - loss_r.item()) < 2e-5
    grads = eng.grads()
    for n, gr in grads_r.items():
        got = grads[n].cpu()
        scale = max(1e-3, gr.abs().max().item())
        assert (got - gr).abs().max().item() < 2e-3 * scale, (n, (got - gr).abs().max().item(), scale)


def test_training_step_bf16_close_to_fp32_oracle(gpu_device):
    """bf16 throughput mode: loss within 1e-2 and gradient directions aligned (cosine > 0.99) with the fp32 oracle"""
    cfg = VarConfig(depth=2)
    vae, m = make(cfg, torch.bfloat16, gpu_device)
    sd = synth_var_state(cfg)
    B, L, fl = 2, cfg.pyramid.L, cfg.pyramid.first_l
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, L - fl, 32, generator=gen)
    tg = torch.randint(0, 4096, (B, L), generator=gen)
    cls, ty = torch.tensor([5, 999]), torch.tensor([1, 3])
    loss_r, _, grads_r = train_ref.loss_and_grads(sd, cfg, cls, x, ty, tg)
    eng = T.TrainEngine(m, drop_path=False)
    loss, _ = eng.forward_backward(cls, x.to(gpu_device), ty, tg.to(gpu_device))
    assert abs(loss.item() - loss_r.item()) < 2e-2
    grads = eng.grads()
    for n, gr in grads_r.items():
        got = grads[n].cpu().flatten().double()
        ref = gr.flatten().double()
        cos = (got @ ref) / (got.norm() * ref.norm() + 1e-30)
        assert cos > 0.99, (n, float(cos))
