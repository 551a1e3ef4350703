import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
for (M, N, K, l) in ((2048, 4608, 1536, 256), (2048, 1536, 1536, 512), (512, 512, 128, 256)):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(T).to(dev); W = (torch.randn(N, K, generator=g) / K ** 0.5).to(T).to(dev)
    b = torch.randn(N, generator=g).to(dev); gate = torch.randn(-(-M // l), N, generator=g).to(dev); x0 = torch.randn(M, N, generator=g).to(dev)
    outs = {}
    for cfg in (2, 28):
        ops.GEMM_TILE_CFG = cfg
        for rep in range(3):
            out = x0.clone()
            ops.gemm(A, W, out, M=M, N=N, K=K, bias=b, gate=gate, ldg=N, gate_rows=l, residual=out, split_k=False)
            torch.cuda.synchronize()
            outs[(cfg, rep)] = out
    ops.GEMM_TILE_CFG = 0
    d = (outs[(2, 0)] != outs[(28, 0)])
    print(M, N, K, 'mismatch', int(d.sum()), 'of', d.numel(), 'rpf run-to-run equal', torch.equal(outs[(2, 0)], outs[(2, 1)]), torch.equal(outs[(2, 1)], outs[(2, 2)]))
    if d.any():
        rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
        print(' rows', rows[:40].tolist(), '... n', len(rows)); print(' rows mod 16 hist', torch.bincount(rows % 16, minlength=16).tolist(), 'rows//16 mod 8 hist', torch.bincount((rows // 16) % 8, minlength=8).tolist())
        print(' cols n', len(cols), 'cols mod 64 hist nonzero', torch.bincount(cols % 64, minlength=64).nonzero().flatten().tolist()[:70], ' col tiles', torch.bincount(cols // 256).tolist())
        r, c = d.nonzero()[0].tolist()
        print(' first', r, c, float(outs[(2, 0)][r, c]), float(outs[(28, 0)][r, c]), 'x0', float(x0[r, c]))
