"""GPU: sampler argument edge cases through the public API - greedy equivalences (top_p -> 0, top_k = 1), no-filter equivalences
(top_k = 0 / V, top_p = 1), determinism under a fixed seed, and the out-of-range top_k error of the reference (torch.topk)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from controlvar_amd import models
dev = torch.device('cuda:0')
vae = models.build_vae(ch=32, compute_dtype=torch.float32).to(dev).eval()
var = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.float32).to(dev).eval()
def ids(**kw):
    var.autoregressive_infer_cfg(B=2, label_B=torch.tensor([3, 7]), cond_type=torch.tensor([0, 1]), cfg=4.0, g_seed=kw.pop('seed', 0), _trace=True, **kw)
    return torch.cat(var.last_trace['idx'], dim=1).cpu()
g = ids(top_k=1, top_p=0.0)
for name, kw in [('top_p=1e-6', dict(top_k=0, top_p=1e-6)), ('top_k=1,top_p=0.5', dict(top_k=1, top_p=0.5)), ('top_k=4096,top_p=1e-7', dict(top_k=4096, top_p=1e-7))]:
    r = ids(**kw)
    print(name, 'equals greedy:', bool(torch.equal(r, g)), int((r != g).sum()))
for name, kw in [('no filter', dict(top_k=0, top_p=0.0)), ('top_k=V', dict(top_k=4096, top_p=0.0)), ('top_p=1.0', dict(top_k=0, top_p=1.0)), ('top_k=4095', dict(top_k=4095, top_p=0.999))]:
    r = ids(**kw); r2 = ids(**kw)
    print(name, 'range ok:', int(r.min()) >= 0 and int(r.max()) < 4096, 'deterministic:', bool(torch.equal(r, r2)), 'differs from greedy:', int((r != g).sum()))
try:
    ids(top_k=5000, top_p=0.0); print('top_k=5000 accepted (the reference raises)')
except Exception as e:
    print('top_k=5000 ->', type(e).__name__, str(e)[:100])
try:
    ids(top_k=-1, top_p=0.0); print('top_k=-1 accepted')
except Exception as e:
    print('top_k=-1 ->', type(e).__name__, str(e)[:100])
