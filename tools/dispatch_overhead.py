#!/usr/bin/env python3
"""Host-side cost per launch of the two bindings of one C-ABI entry point: the ctypes wrapper the model path uses (controlvar_amd/ops.py) and the
torch.library custom op over the same entry (torch.ops.cvar.*, controlvar_amd/torch_ops.py).  A tiny tensor, so the GPU side is negligible and
the loop measures what the host pays per call; also cvar::linear vs ops.gemm (the op allocates its output and checks / reshapes its operands).
Why the model path calls the ctypes layer directly: a d24 generation is ~2 700 launches (10 scales x 24 blocks x 9 + the decoders); at B = 1 it
takes 37-38 ms in total (bench.py side_configs.latency_b1_ms)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import controlvar_amd
from controlvar_amd import ops
ns = controlvar_amd.register_torch_ops()
dev = torch.device('cuda:0')
x = torch.randn(64, 256, device=dev); y = torch.empty(64, 256, device=dev, dtype=torch.bfloat16)
a = torch.randn(64, 256, device=dev).to(torch.bfloat16); w = torch.randn(256, 256, device=dev).to(torch.bfloat16); b = torch.randn(256, device=dev)
o = torch.empty(64, 256, device=dev, dtype=torch.bfloat16)
N = 5000


def per_call(fn):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t1 = time.perf_counter()                      # host time to ENQUEUE N launches (the queue is far from full at these sizes)
    torch.cuda.synchronize()
    return (t1 - t0) / N * 1e6


print(f'silu_cast   ctypes ops.silu_cast        {per_call(lambda: ops.silu_cast(x, y)):6.2f} us per call')
print(f'silu_cast   torch.ops.cvar.silu_cast    {per_call(lambda: ns.silu_cast(x, torch.bfloat16)):6.2f} us per call')
print(f'gemm        ctypes ops.gemm             {per_call(lambda: ops.gemm(a, w, o, M=64, N=256, K=256, bias=b)):6.2f} us per call')
print(f'gemm        torch.ops.cvar.linear       {per_call(lambda: ns.linear(a, w, b, 0)):6.2f} us per call')
