#!/usr/bin/env python3
"""One d24 generation at B = 1 (eager, so that rocprofv3 --kernel-trace sees every launch): warm-up + 3 generations.  usage: b1_profile.py [B=1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=24, mask_type='interleave_append', multi_cond=True).to(dev).eval()
labels = torch.arange(B) % 1000; types = torch.arange(B) % 4
for i in range(4):
    var.autoregressive_infer_cfg(B, labels, g_seed=i, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
torch.cuda.synchronize()
