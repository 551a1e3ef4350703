import torch
dev = torch.device('cuda:0'); T = torch.bfloat16
for (M, N, K) in ((131072, 4608, 1536), (131072, 1536, 6144), (131072, 6144, 1536), (131072, 1536, 1536)):
    A = torch.randn(M, K, device=dev).to(T); W = torch.randn(N, K, device=dev).to(T)
    out = torch.empty(M, N, device=dev, dtype=T)
    for _ in range(3): torch.matmul(A, W.t(), out=out)
torch.cuda.synchronize()
