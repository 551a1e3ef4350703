"""A22 - the data-parallel training runtime: own spawn launcher (replaces mp.spawn + dist.initialize, train_control_var_hpu.py:411-418,
692-697), per-rank data shard (replaces DistributedSampler, :569-574), bucketed gradient all-reduce over engine-shaped slabs (replaces
DDP, :604), and the mean / clip folding of the fused optimizer.  CPU tests run world_size 2 over gloo; the GPU tests put the REAL
training engine's slabs through RCCL in a one-rank group (side stream, async handles, event hand-back)."""
import os

import numpy as np
import pytest
import torch


# ---------------------------------------------------------------------------------------------------------------- sampler (host)
@pytest.mark.parametrize('n,world,drop_last', [(10, 2, False), (11, 2, False), (1001, 8, False), (3, 8, False), (1001, 8, True), (64, 4, True)])
def test_sharded_sampler_equals_torch_distributed_sampler(n, world, drop_last):
    from torch.utils.data import DistributedSampler
    from controlvar_amd.launcher import ShardedSampler
    ds = list(range(n))
    for shuffle in (True, False):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=7, drop_last=drop_last)
            own = ShardedSampler(n, rank=rank, world=world, shuffle=shuffle, seed=7, drop_last=drop_last)
            for epoch in (0, 3):
                ref.set_epoch(epoch); own.set_epoch(epoch)
                assert list(own) == list(ref) and len(own) == len(ref)
            seen += list(own)
        if not drop_last:
            assert set(seen) == set(ds)                         # every item is visited by some rank


def _engine_slab_sizes(depth=2, C=128, V=4096, L=1360, cvae=32, ncls=1000):
    """bucket sizes of train.TrainEngine for a d2 model: [layer slab] * depth + [adaLN generator] + [head + embeddings]"""
    hid = 4 * C
    slab = 3 * C * C + C * C + hid * C + C * hid + 3 * C + C + hid + C + 8
    n_ada = depth * 6 * C + 2 * C
    misc = V * C + V + C * cvae + C + L * C + 10 * C + 2 * C + (ncls + 1) * C + 5 * C
    return [slab] * depth + [n_ada * C + n_ada, misc]


def _dp_worker(rank, world, q):
    """two ranks with different per-rank gradients: bucketed SUM in backward order == the sum computed in one process, and the optimizer-side
    folding (1/world mean, clip coefficient of the MEAN gradient) reproduces single-process clip_grad_norm_ + mean semantics"""
    import torch.distributed as dist
    from controlvar_amd.train import BucketReducer
    sizes = _engine_slab_sizes()
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    per_rank = [[torch.randn(n, generator=g) for n in sizes] for g in gens]         # every rank can rebuild every rank's gradients
    mine = [t.clone() for t in per_rank[rank]]
    red = BucketReducer(mine)
    assert red.world == world and red.active
    for i in range(len(sizes) - 3, -1, -1):          # layers in backward order, then the adaLN generator, then head + embeddings
        red.ready(i)
    red.ready(len(sizes) - 2); red.ready(len(sizes) - 1)
    order = red.wait()
    want = [sum(per_rank[r][i] for r in range(world)) for i in range(len(sizes))]
    ok_sum = all(torch.allclose(a, b, rtol=0, atol=1e-5) for a, b in zip(mine, want))
    # folding: the optimizer sees SUM gradients, scales by 1/world, clips by the norm of the MEAN gradient (train.FusedAdamW.step)
    total_sum = torch.sqrt(sum((t.double() ** 2).sum() for t in mine)).item()
    norm_mean = total_sum / world
    mean = [w / world for w in want]
    ref_norm = torch.sqrt(sum((t.double() ** 2).sum() for t in mean)).item()
    coef = min(1.0, 2.0 / (norm_mean + 1e-6))
    q.put((rank, ok_sum, order, abs(norm_mean - ref_norm) < 1e-6 * ref_norm, coef, red.bytes_sent, sum(sizes) * 4))


def test_engine_shaped_slabs_reduce_over_gloo_world2_via_spawn():
    """uses launcher.spawn (own mp.spawn replacement: env, device, process group on 127.0.0.1) - the same entry a multi-GPU run uses"""
    import torch.multiprocessing as mp
    from controlvar_amd.launcher import spawn
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pc = spawn(_dp_worker, nprocs=2, args=(q,), backend='gloo', port=29300 + os.getpid() % 90, join=False)
    res = sorted(q.get(timeout=180) for _ in range(2))
    import time
    deadline = time.time() + 120
    while not pc.join(timeout=5):                              # join() returns once ONE more process is done; True when all are
        assert time.time() < deadline, 'workers did not exit'
    n = len(_engine_slab_sizes())
    for rank, ok_sum, order, ok_norm, coef, sent, total in res:
        assert ok_sum and ok_norm and 0 < coef <= 1.0
        assert order == list(range(n - 3, -1, -1)) + [n - 2, n - 1]
        assert sent == total                                  # every gradient byte went through the collective exactly once
    assert res[0][4] == res[1][4]                             # both ranks fold the same clip coefficient


def test_synthetic_rank_batches_differ_by_rank():
    from controlvar_amd.launcher import synthetic_rank_batch
    a = synthetic_rank_batch(2, 0, 'cpu', size=32)
    b = synthetic_rank_batch(2, 1, 'cpu', size=32)
    a2 = synthetic_rank_batch(2, 0, 'cpu', size=32)
    assert all(torch.equal(x, y) for x, y in zip(a, a2))
    assert not torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])
    assert a[0].shape == (2, 3, 32, 32) and int(a[2].max()) < 1000 and int(a[3].max()) < 4


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_real_slabs_through_rccl_one_rank_group(gpu_device):
    """Trainer with force_reducer=True inside a 1-rank nccl (= RCCL) group: every per-layer slab of the real engine is all-reduced on the
    reducer's side stream as the backward finishes it; SUM over one rank is the identity, so parameters after two steps must be
    BIT-identical to the un-reduced trainer's - and the reducer must have moved every gradient byte."""
    import torch.distributed as dist
    from controlvar_amd import models, train as T
    from controlvar_amd.synth import synth_images
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29200 + os.getpid() % 90))
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=gpu_device)
        created = True
    try:
        images, masks = synth_images(2, 256, seed=6).to(gpu_device), synth_images(2, 256, seed=7).to(gpu_device)
        cls, types = torch.tensor([17, 403]), torch.tensor([2, 0])
        kw = dict(peak_lr=2e-3, weight_decay=0.05, sche='lin0', warmup_it=2, max_it=50, clip=2.0, drop_path=False)
        states = []
        for force in (False, True):
            vae = models.build_vae(ch=32, compute_dtype=torch.bfloat16).to(gpu_device)
            m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.bfloat16, cond_drop_rate=0.0).to(gpu_device).eval()
            tr = T.Trainer(m, vae, force_reducer=force, **kw)
            outs = [tr.step(images, masks, cls, types) for _ in range(2)]
            torch.cuda.synchronize()
            states.append(({k: v.clone() for k, v in m.state_dict().items()}, [o['loss'].item() for o in outs]))
            if force:
                red = tr.engine.reducer
                assert red is not None and red.active and red.stream is not None
                assert red.bytes_sent == 2 * sum(b.numel() * 4 for b in tr.engine.buckets)
            else:
                assert tr.engine.reducer is None
        assert states[0][1] == states[1][1]
        for k in states[0][0]:
            assert torch.equal(states[0][0][k], states[1][0][k]), k
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_channel_capped_communicators_on_rccl(gpu_device):
    """launcher.channel_groups on the real backend (a one-rank RCCL group is all a single-GPU box offers): every candidate of bench.py's channel-cap
    selection (RCCL's own choice, 16, 8) must yield a communicator that all-reduces correctly, the Trainer must run its bucket exchange through a capped
    group bit-identically to the default one, and pick_fastest must return one of the candidates with a time for each."""
    import torch.distributed as dist
    from controlvar_amd import models, train as T
    from controlvar_amd.launcher import channel_groups, pick_fastest
    from controlvar_amd.synth import synth_images
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29200 + os.getpid() % 90))
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=gpu_device)
        created = True
    try:
        groups = channel_groups((None, 16, 8))
        assert list(groups) == [None, 16, 8] and groups[None] is None and groups[16] is not None and groups[8] is not None
        for c, g in groups.items():
            x = torch.arange(1 << 20, device=gpu_device, dtype=torch.float32)
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=g)
            torch.cuda.synchronize()
            assert torch.equal(x, torch.arange(1 << 20, device=gpu_device, dtype=torch.float32)), c
        images, masks = synth_images(2, 256, seed=6).to(gpu_device), synth_images(2, 256, seed=7).to(gpu_device)
        cls, types = torch.tensor([17, 403]), torch.tensor([2, 0])
        kw = dict(peak_lr=2e-3, weight_decay=0.05, sche='lin0', warmup_it=2, max_it=50, clip=2.0, drop_path=False, force_reducer=True)
        states = []
        for c in (None, 8):
            vae = models.build_vae(ch=32, compute_dtype=torch.bfloat16).to(gpu_device)
            m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.bfloat16, cond_drop_rate=0.0).to(gpu_device).eval()
            tr = T.Trainer(m, vae, **kw)
            tr.set_comm_group(groups[c])
            tr.step(images, masks, cls, types); tr.step(images, masks, cls, types)
            torch.cuda.synchronize()
            assert tr.engine.reducer.group is groups[c]
            states.append({k: v.clone() for k, v in m.state_dict().items()})
        for k in states[0]:
            assert torch.equal(states[0][k], states[1][k]), k
        best, table = pick_fastest([None, 16, 8], lambda c: {None: 0.3, 16: 0.1, 8: 0.2}[c], gpu_device)
        assert best == 16 and list(table) == [None, 16, 8]
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_optimizer_world_folding_on_device(gpu_device):
    """FusedAdamW.step(world=2) on SUM gradients (= 2x the single-rank gradients, what two identical ranks would all-reduce) must land on
    exactly the parameters of world=1 on the single-rank gradients: the 1/world mean and the clip coefficient are folded in-kernel."""
    from controlvar_amd import models, train as T
    gen = torch.Generator().manual_seed(3)
    outs = []
    for world in (1, 2):
        vae = models.build_vae(ch=32, compute_dtype=torch.float32).to(gpu_device)
        m = models.build_control_var(vae, depth=2, mask_type='interleave_append', multi_cond=True, compute_dtype=torch.float32).to(gpu_device)
        opt = T.FusedAdamW(m, lr=1e-2, weight_decay=0.05)
        g0 = torch.Generator().manual_seed(11)
        grads = {n: (torch.randn(p.shape, generator=g0) * 3).to(gpu_device) * world for n, p in m.named_parameters()}
        nc = opt.step(grads, max_norm=2.0, world=world)
        torch.cuda.synchronize()
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, nc.clone().cpu()))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-6)              # grad norm of the mean gradient, clip coefficient
    assert float(outs[0][1][1]) < 1.0                                     # the clip really engaged
    for k in outs[0][0]:
        assert torch.allclose(outs[0][0][k], outs[1][0][k], rtol=0, atol=2e-7), k
