import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops
dev = torch.device('cuda:0'); T = torch.bfloat16
M, N, K = 131072, 6144, 1536
A = torch.randn(M, K, device=dev).to(T); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
out = torch.empty(M, N, device=dev, dtype=T)
for _ in range(4): ops.gemm(A, W, out, M=M, N=N, K=K)
if os.environ.get('WITH_BLAS'):
    for _ in range(4): torch.matmul(A, W.t(), out=out)
torch.cuda.synchronize()
