#!/usr/bin/env python3
"""GPU: drive the public model API through its unusual-but-legal argument forms (reference signatures) and check
shapes / finiteness / determinism.  Not a parity test (tests/ hold those) - a crash-and-garbage detector.

    python tools/api_edge_sweep.py
"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from controlvar_amd import models  # noqa: E402

dev = torch.device('cuda:0')
fails = []


def check(name, fn):
    try:
        out = fn()
        torch.cuda.synchronize()
        if isinstance(out, torch.Tensor):
            assert torch.isfinite(out).all(), 'non-finite output'
        print(f'ok   {name}: {tuple(out.shape) if isinstance(out, torch.Tensor) else out}')
        return out
    except Exception as e:                                          # noqa: BLE001
        fails.append(name)
        print(f'FAIL {name}: {type(e).__name__}: {e}')


for dtype in (torch.bfloat16, torch.float32):
    vae = models.build_vae(ch=32, compute_dtype=dtype).to(dev).eval()
    var = models.build_control_var(vae, depth=4, mask_type='interleave_append', multi_cond=True, compute_dtype=dtype).to(dev).eval()
    pv = models.build_var(vae, depth=4, compute_dtype=dtype).to(dev).eval()
    tag = str(dtype).split('.')[-1]
    for B in (1, 3, 4, 5):
        for label, ctype, cfg, tk, tp, seed in [(None, None if B == 4 else 1, 1.5, 0, 0.0, 1), (7, 2, 0.0, 900, 0.96, None),
                                                 (-1, 3, 4.0, 1, 0.0, 0), (torch.arange(B) * 11, torch.arange(B) % 4, 6, 0, 0.5, 5)]:
            nm = f'{tag} auto B={B} label={label if not isinstance(label, torch.Tensor) else "T"} type={ctype if not isinstance(ctype, torch.Tensor) else "T"} cfg={cfg} k={tk} p={tp} seed={seed}'
            out = check(nm, lambda: var.autoregressive_infer_cfg(B=B, label_B=label, cond_type=ctype, cfg=cfg, top_k=tk, top_p=tp, g_seed=seed))
            if out is not None:
                assert out.shape == (B, 3, 512, 256), out.shape
                assert 0.0 <= float(out.min()) and float(out.max()) <= 1.0
        out1 = check(f'{tag} var B={B}', lambda: pv.autoregressive_infer_cfg(B=B, label_B=None, cfg=1.5, top_k=900, top_p=0.96, g_seed=3))
        out2 = check(f'{tag} var B={B} again', lambda: pv.autoregressive_infer_cfg(B=B, label_B=None, cfg=1.5, top_k=900, top_p=0.96, g_seed=3))
        if out1 is not None and out2 is not None and not torch.equal(out1, out2):
            fails.append(f'{tag} var B={B} not deterministic under a fixed seed'); print('FAIL determinism')
        imgs = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
        ids = check(f'{tag} img_to_idxBl B={B}', lambda: torch.cat(vae.img_to_idxBl(imgs), dim=1))
        idl = vae.img_to_idxBl(imgs)
        for teach in ('c_mask', 'c_img'):
            check(f'{tag} cond B={B} {teach}', lambda: var.conditional_infer_cfg(B=B, label_B=torch.arange(B), cfg=(6, 6, 6), top_k=900, top_p=0.96,
                                                                                  g_seed=2, cond_type=torch.arange(B) % 4, **{teach: idl}))
        check(f'{tag} cond B={B} free', lambda: var.conditional_infer_cfg(B=B, label_B=3, cfg=(1.5, 1.5, 1.5), g_seed=2, cond_type=torch.arange(B) % 4))
        check(f'{tag} idxBl_to_img all scales B={B}', lambda: torch.stack(vae.idxBl_to_img(idl, same_shape=True, last_one=False))[-1])
        check(f'{tag} img_to_recon B={B}', lambda: vae.img_to_recon(imgs, last_one=True))
        x = torch.randn(B, 1358, 32, device=dev)
        with torch.no_grad():
            check(f'{tag} forward B={B}', lambda: var(torch.arange(B, device=dev), x, torch.arange(B, device=dev) % 4))
print('FAILS:', fails)
sys.exit(1 if fails else 0)
