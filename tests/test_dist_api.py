"""controlvar_amd.dist - the reference's `dist.py` helper API (dist.py:15-210) - on two gloo ranks started by the repo's own launcher.spawn,
and in single-process mode (every collective is the identity, as upstream)."""
import torch


def _worker(rank, world, out_dir):
    from controlvar_amd import dist
    assert dist.adopt() and dist.initialized()
    assert (dist.get_rank(), dist.get_world_size()) == (rank, world) and dist.is_master() == (rank == 0)
    t = torch.full((3,), float(rank + 1))
    dist.allreduce(t)
    assert torch.equal(t, torch.full((3,), 3.0))
    g = dist.allgather(torch.tensor([[rank, 10 * rank]]))
    assert g.tolist() == [[0, 0], [1, 10]] and len(dist.allgather(torch.tensor([rank]), cat=False)) == 2
    ragged = dist.allgather_diff_shape(torch.arange(2 + rank, dtype=torch.float32).view(-1, 1) + 100 * rank)
    assert ragged.view(-1).tolist() == [0.0, 1.0, 100.0, 101.0, 102.0]
    b = torch.tensor([7.0 if rank == 1 else 0.0])
    dist.broadcast(b, src_rank=1)
    assert b.item() == 7.0
    assert dist.dist_fmt_vals(rank + 0.5, '%.1f') == ['0.5', '1.5']

    @dist.master_only
    def only_master(x):
        return x * 2

    assert only_master(21) == (42 if rank == 0 else None) and only_master(21, force=True) == 42
    dist.barrier()
    open(f'{out_dir}/ok{rank}', 'w').write('ok')


def test_dist_helpers_world2_gloo(tmp_path):
    from controlvar_amd.launcher import spawn
    spawn(_worker, nprocs=2, args=(str(tmp_path),), backend='gloo')
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def test_dist_helpers_single_process_are_identities():
    from controlvar_amd import dist
    assert not dist.initialized() and dist.get_world_size() == 1 and dist.is_master()
    t = torch.tensor([1.0, 2.0])
    assert dist.allreduce(t) is None and t.tolist() == [1.0, 2.0]
    assert torch.equal(dist.allgather(t), t) and dist.allgather_diff_shape(t, cat=False)[0] is t
    dist.broadcast(t, 0); dist.barrier()
    assert dist.dist_fmt_vals(1.234) == ['1.23']
