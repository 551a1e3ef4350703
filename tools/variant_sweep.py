#!/usr/bin/env python3
"""GPU: the non-default model variants in bf16 throughput mode - generation, conditional generation,
forward and a few fused training steps on a fixed batch (loss must fall).  Crash / garbage detector; parity lives in tests/.

    python tools/variant_sweep.py
"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from controlvar_amd import models, train as T  # noqa: E402
from controlvar_amd.synth import synth_images  # noqa: E402

dev = torch.device('cuda:0')
VARIANTS = {
    'default': {},
    'shared_aln+type_pos': dict(shared_aln=True, type_pos=True),
    'sa_block+layer_scale': dict(aln=-1, layer_scale=0.1),
    'sa_block': dict(aln=-1),
    'bidirectional+type_pos': dict(bidirectional=True, type_pos=True),
    'everything': dict(shared_aln=True, type_pos=True, bidirectional=True),
}
fails = []
for name, kw in VARIANTS.items():
    try:
        vae = models.build_vae(ch=32, compute_dtype=torch.bfloat16).to(dev).eval()
        var = models.build_control_var(vae, depth=4, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.bfloat16,
                                       cond_drop_rate=0.0, **kw).to(dev).eval()
        B = 3
        random.seed(5)
        for _ in range(3):                                   # bidirectional models draw the order per call
            img = var.autoregressive_infer_cfg(B=B, label_B=torch.arange(B), cond_type=torch.arange(B) % 4, cfg=4.0, top_k=900, top_p=0.96, g_seed=1)
            assert img.shape == (B, 3, 512, 256) and torch.isfinite(img).all()
        ids = vae.img_to_idxBl(torch.rand(B, 3, 256, 256, device=dev) * 2 - 1)
        img = var.conditional_infer_cfg(B=B, label_B=torch.arange(B), cfg=(6, 6, 6), top_k=900, top_p=0.96, g_seed=2, cond_type=torch.arange(B) % 4, c_mask=ids)
        assert torch.isfinite(img).all()
        tr = T.Trainer(var, vae, peak_lr=2e-3, weight_decay=0.05, sche='lin0', warmup_it=0, max_it=100, clip=2.0, drop_path=False)
        images, masks = synth_images(2, 256, seed=6).to(dev), synth_images(2, 256, seed=7).to(dev)
        losses = []
        for it in range(6):
            out = tr.step(images, masks, torch.tensor([17, 403]), torch.tensor([2, 0]), mask_first=True)
            losses.append(out['loss'].item())
        assert all(l == l for l in losses) and losses[-1] < losses[0], losses
        if kw.get('bidirectional'):
            out = tr.step(images, masks, torch.tensor([17, 403]), torch.tensor([2, 0]))          # order drawn by python random
            assert out['loss'].item() == out['loss'].item()
        print(f'ok   {name}: loss {losses[0]:.3f} -> {losses[-1]:.3f}')
    except Exception as e:                                    # noqa: BLE001
        import traceback
        traceback.print_exc()
        fails.append(name)
        print(f'FAIL {name}: {type(e).__name__}: {e}')
print('FAILS:', fails)
sys.exit(1 if fails else 0)
