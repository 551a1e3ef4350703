"""One-process-per-GPU launch helpers for the sample-sharded hot path.

Inference and the tokenizer shard by sample: every rank holds a full replica of the weights and runs its own
batch; there is NO data-path collective.  The only communication is the bench/eval harness's barrier and the
max-over-ranks clock (RCCL on GPUs - backend "nccl" is RCCL on ROCm - gloo in the CPU tests).
Replaces the bootstrap of the reference's dist.py:19-48 / train_control_var_hpu.py:411-415 for this path.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Tuple

import torch


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_dist(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed when WORLD_SIZE > 1 (rendezvous on 127.0.0.1 unless MASTER_ADDR is set)."""
    import torch.distributed as dist
    rank, local, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous shard of n_items for this rank (class-range sharding of validate(), train_control_var_hpu.py:366-368)."""
    per = (n_items + world - 1) // world
    return range(min(rank * per, n_items), min((rank + 1) * per, n_items))


def sharded_timed_run(step: Callable[[int], object], steps: int, warmup: int, units_per_step: int,
                      sync: Optional[Callable[[], None]] = None) -> Tuple[float, float]:
    """Run `warmup` untimed + exactly `steps` timed calls of step(i) on every rank, bracketed by barrier + device sync on
    both sides; returns (whole-job units/s = world*units_per_step*steps / max-over-ranks seconds, that max time)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    sync = sync or (lambda: None)

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(warmup):
        step(i)
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return world * units_per_step * steps / dt, dt
