"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange (A22): bucketed SUM all-reduce launched per
finished bucket, mean folded into the optimizer scale."""
import os

import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from controlvar_amd.launcher import init_dist
    from controlvar_amd.train import BucketReducer
    init_dist(backend='gloo')
    buckets = [torch.full((1000 + 7 * i,), float(rank + 1) * (i + 1)) for i in range(5)]
    red = BucketReducer(buckets)
    assert red.world == world
    for i in reversed(range(5)):            # backward order: last layer first
        red.ready(i)
    order = red.wait()
    ok = all(torch.allclose(b, torch.full_like(b, 3.0 * (i + 1))) for i, b in enumerate(buckets))   # (1 + 2) * (i + 1)
    q.put((rank, ok, order))
    dist.destroy_process_group()


def test_bucket_reducer_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, order in res:
        assert ok and order == [4, 3, 2, 1, 0]
