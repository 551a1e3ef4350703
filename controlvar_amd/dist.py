"""The reference's process-group helper API (``dist.py``) over RCCL / gloo - same function names, argument meaning and return shapes, so that
harness code written against ``import dist`` (train_control_var_hpu.py:22, models/control_var.py:11 ``dist.get_device()``) runs unchanged.

Reference (file:line)                      here
  dist.initialize           dist.py:19-48    initialize(): one process per GPU, rendezvous on 127.0.0.1 (launcher.init_dist), RCCL when a GPU is present
  get_rank / get_local_rank / get_world_size / get_device / set_gpu_id / is_master / is_local_master   dist.py:51-83
  new_group                 dist.py:89-92
  barrier                   dist.py:95-97
  allreduce                 dist.py:100-109   SUM, in place; a CPU tensor is reduced through the device when the backend is RCCL
  allgather                 dist.py:112-122   list of per-rank tensors or their concatenation along dim 0
  allgather_diff_shape      dist.py:125-149   ranks may differ in dim 0
  broadcast                 dist.py:152-159
  dist_fmt_vals             dist.py:162-170   one value per rank, formatted
  master_only / local_master_only            dist.py:173-196   run on rank 0 only (``force=True`` overrides), barrier afterwards
  finalize                  dist.py:208-210
Not initialised (single process): every collective is the identity, exactly as upstream.  The models of this package never call these (the
hot path has no data-path collective except the gradient all-reduce, which train.BucketReducer issues itself) - this module is API surface."""
from __future__ import annotations

import functools
import os
from typing import List, Optional, Union

import torch
import torch.distributed as tdist

from .launcher import dist_env, init_dist

_state = {'init': False, 'rank': 0, 'local_rank': 0, 'world': 1, 'device': 'cuda' if torch.cuda.is_available() else 'cpu', 'nccl': False}


def initialized() -> bool:
    return _state['init']


def initialize(fork: bool = False, backend: Optional[str] = None, gpu_id_if_not_distibuted: int = 0, timeout: int = 30):
    """reads RANK / LOCAL_RANK / WORLD_SIZE (torchrun or launcher.spawn); without them: single-process mode on one device"""
    rank, local, world = dist_env()
    if 'WORLD_SIZE' not in os.environ or world <= 1:
        set_gpu_id(gpu_id_if_not_distibuted)
        return
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    init_dist(backend, torch.device('cuda', local % torch.cuda.device_count()) if backend == 'nccl' else None)
    _state.update(init=True, rank=tdist.get_rank(), local_rank=local, world=tdist.get_world_size(), nccl=tdist.get_backend() == 'nccl',
                  device=torch.empty(1, device='cuda').device if backend == 'nccl' else torch.device('cpu'))


def adopt():
    """take over a process group somebody else initialised (launcher.spawn does): fills the rank / world bookkeeping"""
    if tdist.is_available() and tdist.is_initialized():
        nccl = tdist.get_backend() == 'nccl'
        _state.update(init=True, rank=tdist.get_rank(), local_rank=dist_env()[1], world=tdist.get_world_size(), nccl=nccl,
                      device=torch.empty(1, device='cuda').device if nccl else torch.device('cpu'))
    return initialized()


def get_rank() -> int: return _state['rank']
def get_local_rank() -> int: return _state['local_rank']
def get_world_size() -> int: return _state['world']
def get_device(): return _state['device']
def is_master() -> bool: return _state['rank'] == 0
def is_local_master() -> bool: return _state['local_rank'] == 0


def set_gpu_id(gpu_id: Optional[int]):
    if gpu_id is None:
        return
    if not isinstance(gpu_id, (str, int)):
        raise NotImplementedError
    if torch.cuda.is_available():
        torch.cuda.set_device(int(gpu_id))
        _state['device'] = torch.empty(1, device='cuda').device
    else:
        _state['device'] = torch.device('cpu')


def new_group(ranks: List[int]):
    return tdist.new_group(ranks=ranks) if initialized() else None


def barrier():
    if initialized():
        tdist.barrier()


def _on_wire(t: torch.Tensor) -> torch.Tensor:
    """the tensor a collective runs on: RCCL moves device memory only"""
    return t.detach().cuda() if (_state['nccl'] and not t.is_cuda) else t


def allreduce(t: torch.Tensor, async_op: bool = False):
    if not initialized():
        return None
    w = _on_wire(t)
    ret = tdist.all_reduce(w, async_op=async_op)
    if w is not t:
        if async_op:
            ret.wait()
        t.copy_(w.cpu())
    return ret


def allgather(t: torch.Tensor, cat: bool = True) -> Union[List[torch.Tensor], torch.Tensor]:
    if initialized():
        w = _on_wire(t)
        parts = [torch.empty_like(w) for _ in range(get_world_size())]
        tdist.all_gather(parts, w)
    else:
        parts = [t]
    return torch.cat(parts, dim=0) if cat else parts


def allgather_diff_shape(t: torch.Tensor, cat: bool = True) -> Union[List[torch.Tensor], torch.Tensor]:
    if initialized():
        w = _on_wire(t)
        n_here = torch.tensor([w.shape[0]], device=w.device, dtype=torch.int64)
        counts = [torch.empty_like(n_here) for _ in range(get_world_size())]
        tdist.all_gather(counts, n_here)
        counts = [int(c.item()) for c in counts]
        longest = max(counts)
        if longest > w.shape[0]:
            w = torch.cat((w, w.new_empty((longest - w.shape[0],) + tuple(w.shape[1:]))), dim=0)
        padded = [torch.empty_like(w) for _ in counts]
        tdist.all_gather(padded, w.contiguous())
        parts = [p[:n] for p, n in zip(padded, counts)]
    else:
        parts = [t]
    return torch.cat(parts, dim=0) if cat else parts


def broadcast(t: torch.Tensor, src_rank: int) -> None:
    if not initialized():
        return
    w = _on_wire(t)
    tdist.broadcast(w, src=src_rank)
    if w is not t:
        t.copy_(w.cpu())


def dist_fmt_vals(val: float, fmt: Optional[str] = '%.2f'):
    if not initialized():
        return torch.tensor([val]) if fmt is None else [fmt % val]
    per_rank = torch.zeros(get_world_size())
    per_rank[get_rank()] = val
    allreduce(per_rank)
    return per_rank if fmt is None else [fmt % v for v in per_rank.tolist()]


def _only(pred):
    def deco(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            force = kwargs.pop('force', False)
            out = func(*args, **kwargs) if (force or pred()) else None
            barrier()
            return out
        return wrapper
    return deco


master_only = _only(is_master)
local_master_only = _only(is_local_master)


def finalize():
    if initialized():
        tdist.destroy_process_group()
        _state.update(init=False, rank=0, local_rank=0, world=1, nccl=False)
