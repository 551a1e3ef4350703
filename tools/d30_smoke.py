#!/usr/bin/env python3
"""BASELINE config 4: d30 (cos-attn) inference, B=4, cond_type=None -> [mask, canny, depth, normal], cfg=4; also conditional_infer_cfg."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
from controlvar_amd.synth import synth_images
dev = torch.device('cuda:0')
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=30, mask_type='interleave_append', multi_cond=True).to(dev).eval()
labels = torch.tensor([1, 2, 3, 4])
img = var.autoregressive_infer_cfg(4, labels, g_seed=0, cfg=4.0, top_k=900, top_p=0.96, cond_type=None)
torch.cuda.synchronize(); t0 = time.perf_counter()
img = var.autoregressive_infer_cfg(4, labels, g_seed=1, cfg=4.0, top_k=900, top_p=0.96, cond_type=None)
torch.cuda.synchronize(); t1 = time.perf_counter(); t_gen = t1 - t0
c_ids = vae.img_to_idxBl(synth_images(4, 256, seed=9).to(dev))      # warm-up (first use loads the encoder's kernels)
var.conditional_infer_cfg(4, labels, g_seed=0, cfg=(4.0, 4.0, 4.0), top_k=900, top_p=0.96, cond_type=torch.tensor([0, 1, 2, 3]), c_mask=c_ids)
torch.cuda.synchronize(); t1 = time.perf_counter()
c_ids = vae.img_to_idxBl(synth_images(4, 256, seed=9).to(dev))
img2 = var.conditional_infer_cfg(4, labels, g_seed=1, cfg=(4.0, 4.0, 4.0), top_k=900, top_p=0.96, cond_type=torch.tensor([0, 1, 2, 3]), c_mask=c_ids)
torch.cuda.synchronize(); t2 = time.perf_counter()
B = 64
lab = torch.arange(B) % 1000; ty = torch.arange(B) % 4
var.autoregressive_infer_cfg(B, lab, g_seed=2, cfg=4.0, top_k=900, top_p=0.96, cond_type=ty); torch.cuda.synchronize(); t3 = time.perf_counter()
var.autoregressive_infer_cfg(B, lab, g_seed=3, cfg=4.0, top_k=900, top_p=0.96, cond_type=ty); torch.cuda.synchronize(); t4 = time.perf_counter()
print(json.dumps({'d30_B4_all_cond_types_s': round(t_gen, 3), 'img_shape': list(img.shape), 'finite': bool(torch.isfinite(img).all()),
                  'conditional_infer_cfg_B4_s': round(t2 - t1, 3), 'd30_B64_images_per_s': round(B / (t4 - t3), 1)}))
