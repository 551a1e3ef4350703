#!/usr/bin/env python3
"""ms_encode alone (the ten-scale residual quantiser of the tokenizer, quant.py:184-215): ms per call at a few batch sizes, ids of the fast search against the
sequential search (margin path).  A/B of search forms: CVAR_LIB=ab/libcvar_<tag>.so.  usage: ms_encode_bench.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import models
dev = torch.device('cuda:0')
vae = models.build_vae(ch=160).to(dev)
vae._pack()
print('lib', os.environ.get('CVAR_LIB', 'default'))
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 128, 256]:
    f = (torch.randn(B, 32, 16, 16, generator=torch.Generator().manual_seed(B)) * 0.7).to(dev)
    idx, _, _ = vae._ms_encode(f)
    ref, _, _ = vae._ms_encode(f, want_margin=True)                     # sequential search
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        vae._ms_encode(f)
    e1.record(); torch.cuda.synchronize()
    print(f'B={B:4d}  {e0.elapsed_time(e1) / 5:7.3f} ms per call   ids equal to the sequential search: {bool(torch.equal(idx, ref))}  ({int((idx != ref).sum())} of {idx.numel()} differ)', flush=True)
