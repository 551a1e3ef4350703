"""world_size-2 gloo test (CPU) of the sample-sharded launcher used by bench.py for N > 1."""
import os
import time

import pytest
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from controlvar_amd.launcher import init_dist, shard_range, sharded_timed_run
    import torch.distributed as dist
    r, l, w = init_dist(backend='gloo')
    assert (r, w) == (rank, world)
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.05 if rank == 0 else 0.15)          # rank 1 is the slow one

    value, dt = sharded_timed_run(step, steps=3, warmup=1, units_per_step=8)
    # every rank reports the same max-over-ranks clock
    t = torch.tensor([dt], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    q.put((rank, value, dt, len(calls), float(lo), float(hi), list(shard_range(1000, rank, world))[:1] + [len(shard_range(1000, rank, world))]))
    dist.destroy_process_group()


def test_sharded_timed_run_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, value, dt, ncalls, lo, hi, shard in res:
        assert ncalls == 4                                   # 1 warm-up + exactly 3 timed steps
        assert abs(lo - hi) < 1e-9                           # identical clock on all ranks
        assert 0.44 < dt < 1.5                               # >= 3 x 0.15 s: the slow rank sets the time
        assert abs(value - 2 * 8 * 3 / dt) < 1e-9            # whole-job aggregate, weak scaling
    assert res[0][6] == [0, 500] and res[1][6] == [500, 500]
