#!/usr/bin/env python3
"""BASELINE config 3 on one GPU: d24 joint image+control training step, synthetic ImageNetC-shaped batch (B per GPU)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models, train as T
from controlvar_amd.launcher import dist_env, init_dist
from controlvar_amd.synth import synth_images
from controlvar_amd.spec import VarConfig, algorithmic_gflop_per_row, VAE_ENCODE_GFLOP

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rank, local, world = dist_env()
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
init_dist('nccl', dev)
vae = models.build_vae(ch=160).to(dev)
var = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True).to(dev).train()
tr = T.Trainer(var, vae, peak_lr=8e-5 * (B * world) / 512, weight_decay=0.08, sche='lin0', warmup_it=10, max_it=1000, clip=2.0)
images, masks = synth_images(B, 256, seed=rank).to(dev), synth_images(B, 256, seed=100 + rank).to(dev)
cls = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(rank)); types = torch.arange(B) % 4
out = tr.step(images, masks, cls, types)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.step(images, masks, cls, types)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
if rank == 0:
    fl = algorithmic_gflop_per_row(VarConfig(depth=depth), n_ada=1)
    per_sample = 3 * fl['total'] + 2 * VAE_ENCODE_GFLOP
    print(json.dumps({'metric': f'd{depth} training samples/s', 'value': round(B * world / dt, 2), 'ms_per_step': round(dt * 1e3, 1), 'batch_per_gpu': B,
                      'world': world, 'loss': round(float(out['loss']), 4), 'grad_norm': round(float(out['grad_norm']), 4),
                      'algorithmic_tflops_per_gpu': round(per_sample * B / dt / 1e3, 1), 'mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
