"""One-process-per-GPU launch helpers for the data-parallel hot path (SURVEY.md sections 8a A22 and 8e).

Inference and the tokenizer shard by sample: every rank holds a full replica of the weights and runs its own
batch; there is NO data-path collective.  The only communication is the bench/eval harness's barrier and the
max-over-ranks clock (RCCL on GPUs - backend "nccl" is RCCL on ROCm - gloo in the CPU tests).
Replaces the bootstrap of the reference's dist.py:19-48 / train_control_var_hpu.py:411-415 for this path.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Sequence, Tuple

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
    sync()
    own = time.perf_counter() - t0                         # this rank's own K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    LAST_RUN['rank_seconds'] = [own]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        o = torch.tensor([own], dtype=torch.float64)
        if dist.get_backend() == 'nccl':
            t, o = t.cuda(), o.cuda()
        every = [torch.zeros_like(o) for _ in range(world)]
        dist.all_gather(every, o)                          # every rank's own clock (reported per rank, see collective_evidence)
        LAST_RUN['rank_seconds'] = [float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return world * units_per_step * steps / dt, dt


LAST_RUN: dict = {'rank_seconds': []}                       # per-rank seconds of the last sharded_timed_run (rank order)


def collective_evidence(device: Optional[torch.device] = None) -> dict:
    """What proves "the collective library saw N ranks" from inside the job: a REAL all_reduce of one element of ones over the
    default group (SUM == number of ranks that took part), the backend and its library version.  World 1 without a group: one rank, no call."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return {'collective_backend': None, 'rccl_ranks': 1, 'rccl_version': None}
    backend = dist.get_backend()
    one = torch.ones(1, dtype=torch.float32, device=device if (backend == 'nccl' and device is not None) else 'cpu')
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    ver = None
    if backend == 'nccl':
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                   # never lose the bench line over a version string
            ver = 'unknown'
    return {'collective_backend': 'rccl' if backend == 'nccl' else backend, 'rccl_ranks': int(round(float(one.item()))), 'rccl_version': ver}


def channel_groups(candidates: Sequence[Optional[int]] = (None, 16, 8)):
    """One process group per RCCL channel cap (``ncclConfig_t.maxCTAs`` through ProcessGroupNCCL.Options - a PER-COMMUNICATOR setting, unlike
    NCCL_MAX_NCHANNELS, which RCCL reads once per process): None = the default group (RCCL's own choice).  An all-reduce kernel occupies one
    workgroup per channel; the backward GEMMs it overlaps with are sized to exactly 256 CUs (gemm.hip tile plan), so how many CUs the exchange
    takes is a real trade the first 8-GPU run has to settle by measurement.  Under gloo (CPU tests of the
    selection logic) every candidate is a plain new group.  Every rank must call this with the same candidates."""
    import torch.distributed as dist
    out = {}
    for c in candidates:
        if c is None or not dist.is_initialized():
            out[c] = None
        elif dist.get_backend() == 'nccl':
            opts = dist.ProcessGroupNCCL.Options()
            opts.config.max_ctas = int(c)
            opts.config.min_ctas = min(int(c), 1)
            out[c] = dist.new_group(pg_options=opts)
        else:
            out[c] = dist.new_group()
    return out


def pick_fastest(labels: Sequence, seconds_of: Callable[[object], float], device: Optional[torch.device] = None):
    """Run ``seconds_of(label)`` for every label on every rank, agree on the MAX over ranks of each (one all-reduce, so that all ranks take
    the same decision from the same numbers) and return (label with the smallest time - the first on ties -, {label: seconds})."""
    import torch.distributed as dist
    local = [float(seconds_of(lb)) for lb in labels]
    t = torch.tensor(local, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == 'nccl':
            t = t.to(device if device is not None else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    agreed = [float(x) for x in t.cpu()]
    best = min(range(len(labels)), key=lambda i: (agreed[i], i))
    return labels[best], {labels[i]: agreed[i] for i in range(len(labels))}


# ---------------------------------------------------------------------------------------------------------------------------------
# A22: the data-parallel training runtime around the step (train_control_var_hpu.py:411-418,569-574,604,692-697; dist.py:19-48)
# ---------------------------------------------------------------------------------------------------------------------------------
def _spawn_entry(rank: int, fn, world: int, port: int, backend: Optional[str], args: tuple):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')            # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    import torch.distributed as dist
    init_dist(backend, torch.device('cuda', rank % torch.cuda.device_count()) if torch.cuda.is_available() and backend != 'gloo' else None)
    try:
        fn(rank, world, *args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def spawn(fn: Callable, nprocs: Optional[int] = None, args: tuple = (), backend: Optional[str] = None, port: Optional[int] = None, join: bool = True):
    """``mp.spawn(main_worker, nprocs=ngpus_per_node, args=...)`` of the reference (train_control_var_hpu.py:692-697) without torchrun:
    starts `nprocs` processes (default: one per visible GPU), each with RANK / LOCAL_RANK / WORLD_SIZE set, the device selected and
    torch.distributed initialised on 127.0.0.1 (RCCL when GPUs are present, gloo otherwise), and calls ``fn(rank, world, *args)`` in it.
    The same worker also runs under torchrun: init_dist() reads the environment either way."""
    import torch.multiprocessing as mp
    if nprocs is None:
        nprocs = max(1, torch.cuda.device_count()) if torch.cuda.is_available() else 1
    if port is None:
        port = 29400 + (os.getpid() * 7) % 500
    return mp.start_processes(_spawn_entry, args=(fn, nprocs, port, backend, tuple(args)), nprocs=nprocs, join=join, start_method='spawn')


class ShardedSampler:
    """Per-rank index shard of a data set - the arithmetic of ``torch.utils.data.DistributedSampler`` as the reference uses it
    (train_control_var_hpu.py:569-574: shuffle=True, drop_last=False; ``set_epoch`` per epoch :651): epoch permutation from
    ``seed + epoch``, padded by wrapping to a multiple of the world size, rank r takes every world-th index starting at r.
    Stateless apart from the epoch, so every rank derives its shard without communication."""

    def __init__(self, n_items: int, rank: Optional[int] = None, world: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        r, _, w = dist_env()
        self.n, self.rank, self.world = int(n_items), r if rank is None else int(rank), w if world is None else int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError(f'rank {self.rank} outside world {self.world}')
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, 0
        if drop_last and self.n % self.world:
            self.num_samples = self.n // self.world
        else:
            self.num_samples = (self.n + self.world - 1) // self.world
        self.total_size = self.num_samples * self.world

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad > 0:
                idx += (idx * ((pad + len(idx) - 1) // max(len(idx), 1) + 1))[:pad]
        else:
            idx = idx[:self.total_size]
        return iter(idx[self.rank:self.total_size:self.world])


def synthetic_rank_batch(batch: int, rank: int, device, size: int = 256):
    """the synthetic ImageNetC-shaped batch of one rank - image / control U-shaped fields seeded by the rank,
    class ids ~ U{0..999}, condition types ~ U{0..3} (deterministic per rank, different across ranks)."""
    from .synth import synth_images
    g = torch.Generator().manual_seed(1000 + rank)
    return (synth_images(batch, size, seed=2 * rank).to(device), synth_images(batch, size, seed=2 * rank + 1).to(device),
            torch.randint(0, 1000, (batch,), generator=g), torch.randint(0, 4, (batch,), generator=g))
