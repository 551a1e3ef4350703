import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, ops
dev = torch.device('cuda:0')
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=24, mask_type='interleave_append', multi_cond=True).to(dev).eval()
for rep in range(2):
    res = {}
    for B in (1, 4, 16):
        labels = torch.arange(B) % 1000; types = torch.arange(B) % 4
        run = var.graphed_generator(B, cfg=4.0, top_k=900, top_p=0.96)
        run(labels, types, g_seed=0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5): out = run(labels, types, g_seed=i)
        torch.cuda.synchronize(); res[B] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
        del run
    print('graph ms per generation', res, flush=True)
