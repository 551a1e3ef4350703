#!/usr/bin/env python3
"""BASELINE config 5: VQVAE-only encode -> multi-scale quant -> decode throughput, 256^2, B images per GPU."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
from controlvar_amd.synth import synth_images
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda:0')
models.VQVAE.GN_FROM_CONV = os.environ.get('GN_FROM_CONV', '1') != '0'          # A/B: GroupNorm statistics from the conv epilogue (round 5) or the stand-alone pass
vae = models.build_vae(ch=160, decode_chunk=int(os.environ.get('DECODE_CHUNK', '128'))).to(dev)          # images per decoder pass
img = synth_images(B, 256, seed=3).to(dev)
def step():
    CH = int(os.environ.get('VAE_CHUNK', '128'))          # images per encode + quantise call (round 4: 128, was 64; the decoder chunks by itself)
    outs = []
    for s in range(0, B, CH):
        ids = vae.img_to_idxBl(img[s:s + CH])
        outs.append(vae.idxBl_to_img(ids, same_shape=True, last_one=True))
    return outs
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 3
for _ in range(n): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps({'gn_from_conv': models.VQVAE.GN_FROM_CONV, 'metric': 'VQVAE encode+quant+decode images/s (256^2)', 'value': round(B / dt, 1), 'batch': B, 'ms_per_pass': round(dt * 1e3, 1),
                  'algorithmic_tflops': round(609e9 * B / dt / 1e12, 1), 'mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
