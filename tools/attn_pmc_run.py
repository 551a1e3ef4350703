import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
from controlvar_amd.spec import VarConfig
dev = torch.device('cuda:0'); T = torch.bfloat16
cfg = VarConfig(depth=24); py = cfg.pyramid
R, H, L, C = 256, cfg.H, py.L, cfg.C
qkv = (torch.randn(R, L, 3 * C, device=dev) * 0.5).to(T)
b, e = py.begin[9], py.end[9]; l = e - b
out = torch.empty(R * l, C, device=dev, dtype=T)
for _ in range(3): ops.attention(qkv, out, R, H, L, b, l, 0.03125, None)
torch.cuda.synchronize()
