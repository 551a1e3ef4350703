#!/usr/bin/env python3
"""Headline benchmark: images/s of 256^2 ControlVAR d24 `autoregressive_infer_cfg` (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one full generation pass over a synthetic batch of B class/condition labels per GPU:
10 coarse-to-fine scales x depth blocks with the multi-scale KV arena, 2-way CFG, top-k/top-p
sampling over the 4096-way codebook, token->feature pyramid, and BOTH VQVAE decodes (control + image).
Inputs (labels, condition types, synthetic weights) are resident in HBM before the timed region.
Inference shards by sample: no data-path collective ("scaling": "weak").

The JSON line also carries
  roofline     - the dominant kernel (the MFMA GEMM / implicit-conv kernel): algorithmic FLOPs of its launches
                 divided by their HIP-event-timed duration inside the timed region, against the 2.5 PFLOP/s bf16 peak;
  cpu_baseline - the CPU oracle (this repo's restatement of the reference, validated against it) timed on the
                 host cores of rank 0 on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--depth', type=int, default=24)
    ap.add_argument('--batch', type=int, default=0, help='samples per GPU per step; 0 = 384 up to d24, 128 above (K/V arena: 0.4 GB per sample at d24 bf16 '
                                                          '-> 154 GB at 384, peak allocation 172 GB of the 288 GB; larger batches fill the partial tile rounds of the '
                                                          'mid scales: 128 -> 256 +2.5 %%, 256 -> 384 +0.8 %%, 384 -> 512 +0.7 %% at 226 GB)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cfg', type=float, default=4.0)
    ap.add_argument('--top_k', type=int, default=900)      # the reference's sampling defaults (train_control_var_hpu.py:338)
    ap.add_argument('--top_p', type=float, default=0.96)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-depth', type=int, default=0, help='depth of the CPU baseline model (0 = same as --depth)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="infer (default, the BASELINE headline metric) | train: BASELINE config 3 - the d24 data-parallel training step, gradient "
                         "all-reduce over RCCL overlapped with the backward; reports samples/s and the exposed communication share")
    ap.add_argument('--train-batch', type=int, default=32, help='--mode train: samples per GPU per step (global 256 = 8 x 32)')
    return ap.parse_args()


def physical_cores() -> int:
    """physical cores of the host (BASELINE.md section 4 asks for the CPU path on the physical cores): unique (package, core) pairs of
    /proc/cpuinfo restricted to the CPUs this process may run on; falls back to the affinity count"""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    try:
        cores, cur = set(), {}
        for line in open('/proc/cpuinfo'):
            if ':' in line:
                k, v = (x.strip() for x in line.split(':', 1))
                cur[k] = v
            elif not line.strip() and cur:
                if int(cur.get('processor', -1)) in allowed:
                    cores.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
                cur = {}
        if cur and int(cur.get('processor', -1)) in allowed:
            cores.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
        if cores:
            return len(cores)
    except Exception:
        pass
    return max(1, len(allowed))


def cpu_baseline(depth: int, seed: int = 0, budget_s: float = 15.0):
    """The oracle (kind 'port': own restatement, pinned to the reference by tests/golden) on the host cores.
    Sample: ONE d{depth} generation with B=1 (2 CFG rows), fp32, greedy - run scale by scale until `budget_s` seconds
    are spent; the rate is extrapolated by the share of the sample's algorithmic FLOPs completed (both VAE decodes
    are included only if every scale finished inside the budget).  Timed with one thread per PHYSICAL core
    and, when the host has more than 32 of them, also with 32 threads - torch's CPU GEMMs at B=1 do not scale to 128 threads, and
    the faster of the two is what is reported (`cores` = the threads of the reported run; both are named in `sample`)."""
    from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VAE_DECODE_GFLOP, VaeConfig, VarConfig, phi_index_map
    from controlvar_amd.synth import synth_vae_state, synth_var_state
    from oracle import var_ref
    from oracle.vqvae_ref import MSQuant
    cfg = VarConfig(depth=depth)
    py, C, V = cfg.pyramid, cfg.C, cfg.vocab
    per_scale = [2 * (depth * (24 * C * C * l + 4 * C * l * e) + 2 * C * V * l) / 1e9 for l, e in zip(py.l, py.end)]   # 2 CFG rows
    total = sum(per_scale) + 2 * VAE_DECODE_GFLOP
    sdv = synth_vae_state(VaeConfig(ch=160), seed)
    sd = synth_var_state(cfg, seed)
    msq = MSQuant(sdv, PN, phi_index_map(10))

    def run(threads):
        torch.set_num_threads(threads)
        done = {'n': 0}
        t0 = time.time()

        def hook(si):
            done['n'] = si + 1
            return (time.time() - t0) > budget_s

        with torch.no_grad():
            f = var_ref.generate(sd, cfg, msq, 1, torch.tensor([7]), 4.0, top_k=1, cond_type=torch.tensor([1]), stage_hook=hook)
            gf = sum(per_scale[:done['n']])
            if done['n'] == len(per_scale):
                var_ref.decode_fhat(sdv, f)
                gf += 2 * VAE_DECODE_GFLOP
        dt = time.time() - t0
        return dict(rate=(gf / total) / dt, threads=threads, scales=done['n'], share=gf / total, dt=dt)

    phys = physical_cores()
    runs = [run(phys)] + ([run(32)] if phys > 32 else [])
    best = max(runs, key=lambda r: r['rate'])
    desc = '; '.join(f'{r["threads"]} threads: {r["scales"]}/10 scales{" + 2 VAE decodes" if r["scales"] == 10 else ""} = {100 * r["share"]:.1f}% of the per-image FLOPs in '
                     f'{r["dt"]:.1f}s -> {r["rate"]:.4f} img/s' for r in runs)
    return dict(value=best['rate'], unit='images/s', cores=best['threads'], kind='port',
                sample=f'd{depth} B=1 (2 CFG rows) fp32 torch-CPU oracle, greedy, rate extrapolated by FLOP share; host has {phys} physical cores; {desc}')


def main_train(a):
    """BASELINE config 3: d24 joint image+control training, synthetic ImageNetC-shaped batch of --train-batch samples per GPU (seeded by
    the rank), frozen tokenizer inside the step, per-layer gradient slabs all-reduced (SUM, mean folded into AdamW) over RCCL on a side
    stream while the backward continues.  The same K steps are timed once more with the exchange switched off; the difference is the
    EXPOSED communication time (what the overlap did not hide)."""
    from controlvar_amd.launcher import dist_env, init_dist, sharded_timed_run, synthetic_rank_batch
    rank, local, world = dist_env()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    init_dist('nccl', dev)
    from controlvar_amd import models, ops, train as T, _lib
    from controlvar_amd.spec import VarConfig, algorithmic_gflop_per_row
    _lib.load()
    B = a.train_batch
    Tt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    vae = models.build_vae(ch=160, compute_dtype=Tt).to(dev)
    var = models.build_control_var(vae, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=Tt).to(dev)
    tr = T.Trainer(var, vae, peak_lr=8e-5 * (B * world) / 512, weight_decay=0.08, sche='lin0', warmup_it=10, max_it=10000, clip=2.0, train_mode=True)
    images, masks, cls, types = synthetic_rank_batch(B, rank, dev)
    last = {}

    def step(i):
        last['out'] = tr.step(images, masks, cls, types, drop_seed=1000 * rank + i)

    _, dt = sharded_timed_run(step, a.steps, a.warmup, B, sync=torch.cuda.synchronize)
    exposed = None
    if world > 1:
        tr.comm = False
        _, dt_nocomm = sharded_timed_run(step, a.steps, 1, B, sync=torch.cuda.synchronize)
        tr.comm = True
        exposed = max(0.0, (dt - dt_nocomm) / dt)
    if rank == 0:
        fl = algorithmic_gflop_per_row(VarConfig(depth=a.depth), n_ada=1)
        per_sample_tf = (3 * fl['total'] + 2 * 215.4) / 1e3                   # fwd + 2x bwd + two frozen tokenizer encodes
        red = tr.engine.reducer if tr.engine.reducer is not None else getattr(tr, '_reducer', None)
        out = {'metric': 'training samples/sec (d%d joint image+control, DP all-reduce)' % a.depth, 'value': round(world * B * a.steps / dt, 2),
               'unit': 'samples/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': f'ControlVAR d{a.depth} training step (tokenise image+control, forward, CE, backward, clip 2.0, AdamW), '
                                      f'synthetic ImageNetC-shaped batch', 'batch_per_gpu': B, 'global_batch': B * world, 'seq_len': 1360,
                          'parallelism': f'dp{world} (per-layer gradient slabs, RCCL all-reduce on a side stream)'},
               'algorithmic_tflop_per_sample': round(per_sample_tf, 3), 'end_to_end_tflops_per_gpu': round(per_sample_tf * B * a.steps / dt, 1),
               'loss': round(float(last['out']['loss']), 4), 'exposed_comm_frac': None if exposed is None else round(exposed, 4),
               'allreduce_bytes_per_step': sum(b.numel() * 4 for b in tr.engine.buckets)}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.mode == 'train':
        return main_train(a)
    from controlvar_amd.launcher import dist_env, init_dist, sharded_timed_run
    rank, local, world = dist_env()
    torch.cuda.set_device(local)
    init_dist('nccl', torch.device('cuda', local))
    if a.gpus != world and world > 1 and rank == 0:
        print(f'[bench] --gpus {a.gpus} != WORLD_SIZE {world}; using WORLD_SIZE', file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from controlvar_amd import models, ops, _lib
    from controlvar_amd.spec import VarConfig, algorithmic_gflop_per_row, VAE_DECODE_GFLOP
    _lib.load()
    T = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    t_build = time.time()
    vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
    var = models.build_control_var(vae, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
    B = a.batch or (384 if a.depth <= 24 else 128)
    g = torch.Generator().manual_seed(1234 + rank)
    labels = torch.randint(0, 1000, (B,), generator=g).to(dev)
    types = (torch.arange(B) % 4).to(dev)
    var._pack(); vae._pack()
    torch.cuda.synchronize()
    if rank == 0:
        print(f'[bench] built d{a.depth} + VQVAE ch160 ({a.dtype}) in {time.time() - t_build:.1f}s; B={B}/GPU, world={world}', file=sys.stderr)

    def step(seed):
        return var.autoregressive_infer_cfg(B, labels, g_seed=seed, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p, cond_type=types)

    prof = None if a.no_kernel_timing else []
    last = {}

    def timed_step(i):
        if i == a.warmup:
            ops.GEMM_PROFILE = prof                     # kernel events only inside the timed region
        last['img'] = step(100 + i)

    _, dt = sharded_timed_run(timed_step, a.steps, a.warmup, B, sync=torch.cuda.synchronize)
    ops.GEMM_PROFILE = None
    img = last['img']
    assert img.shape == (B, 3, 512, 256)

    if rank == 0:
        cfg = VarConfig(depth=a.depth)
        fl = algorithmic_gflop_per_row(cfg, n_ada=1)
        per_sample_tf = (2 * fl['total'] + 2 * VAE_DECODE_GFLOP) / 1e3
        value = world * B * a.steps / dt
        out = {
            'metric': 'images/sec (256^2 autoregressive_infer_cfg, d%d)' % a.depth, 'value': round(value, 3), 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'ControlVAR d{a.depth} autoregressive_infer_cfg 256^2 (control+image), CFG={a.cfg}, '
                                   f'top_k={a.top_k}, top_p={a.top_p}, incl. 2 VQVAE decodes; synthetic weights/labels',
                       'batch_per_gpu': B, 'global_batch': B * world, 'seq_len': cfg.pyramid.L, 'parallelism': f'dp{world} (sample-sharded, no collective)'},
            'algorithmic_tflop_per_image': round(per_sample_tf, 3),
            'end_to_end_tflops_per_gpu': round(per_sample_tf * B * a.steps / dt, 1),
            'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
        }
        if prof:
            ms = sum(r[0].elapsed_time(r[1]) for r in prof)
            flops = sum(r[2] for r in prof)
            ach = flops / (ms * 1e-3) / 1e12
            peak = 2500.0 if a.dtype == 'bf16' else 157.3
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'gemm_hbm_traffic.json')
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get('bytes_per_launch')
                except Exception:
                    traffic = None
            out['roofline'] = {'bound': 'mfma', 'kernel': 'cvar_gemm_kernel + conv3x3_halo_bf16_kernel (every cvar_gemm launch: GEMMs and 3x3 convs)',
                               'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                               'traffic': traffic, 'launches': len(prof), 'avg_launch_ms': round(ms / len(prof), 4),
                               'gemm_share_of_step': round(ms * 1e-3 / dt, 3)}
        if not a.no_cpu_baseline and world == 1:            # reported baseline: rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(a.cpu_depth or a.depth)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
