#!/usr/bin/env python3
"""Small-batch latency: eager launch sequence vs the captured HIP graph (d24)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, ops
ops.GEMM_TILE_CFG = int(os.environ.get('ISO_CFG', '0'))
ops.SMALL_M_KERNEL = os.environ.get('SMALLM', '1') != '0'        # 0: the transformer's small passes stay on the LDS-tiled kernels + split-K      # 6: every eligible 3x3 conv on the LDS-halo kernel whatever the grid size
models.FUSE_LN_BELOW = int(os.environ.get('FUSE_LN_BELOW', models.FUSE_LN_BELOW))
dev = torch.device('cuda:0')
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True).to(dev).eval()
res = {}
for B in [int(b) for b in os.environ.get('LAT_B', '1,4,8,16').split(',')]:
    labels = torch.arange(B) % 1000; types = torch.arange(B) % 4
    var.autoregressive_infer_cfg(B, labels, g_seed=0, cfg=4.0, top_k=900, top_p=0.96, cond_type=types); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3): var.autoregressive_infer_cfg(B, labels, g_seed=i, cfg=4.0, top_k=900, top_p=0.96, cond_type=types)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 3
    run = var.graphed_generator(B, cfg=4.0, top_k=900, top_p=0.96)
    run(labels, types, g_seed=0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3): run(labels, types, g_seed=i)
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 3
    res[f'B={B}'] = dict(eager_ms=round(eager * 1e3, 1), graph_ms=round(graph * 1e3, 1), graph_img_s=round(B / graph, 1))
    del run
print(json.dumps(res))
