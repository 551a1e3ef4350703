#!/usr/bin/env python3
"""Headline benchmark: images/s of 256^2 ControlVAR d24 `autoregressive_infer_cfg` (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the
environment: this process IS one rank), and as plain `python bench.py --gpus N` - then bench.py starts the N ranks itself with
controlvar_amd.launcher.spawn, the counterpart of the reference's `mp.spawn(main_worker, nprocs=ngpus_per_node)`
(train_control_var_hpu.py:692-697): one process per GPU, RCCL group on 127.0.0.1, rank 0 prints the one JSON line.

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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--depth', type=int, default=24)
    ap.add_argument('--batch', type=int, default=0, help='samples per GPU per step; 0 = 512 up to d24, 128 above (K/V arena: 0.4 GB per sample at d24 bf16 '
                                                          '-> 205 GB at 512, peak allocation 226 GB of the 288 GB; larger batches fill the partial tile rounds of the '
                                                          'mid scales: 128 -> 256 +2.5 %%, 256 -> 384 +0.8 %%, 384 -> 512 +0.7 %% at 226 GB)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cfg', type=float, default=4.0)
    ap.add_argument('--top_k', type=int, default=900)      # the reference's sampling defaults (train_control_var_hpu.py:338)
    ap.add_argument('--top_p', type=float, default=0.96)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--decode-chunk', type=int, default=0, help='images per VQVAE decoder pass (0 = the model default); A/B knob')
    ap.add_argument('--cpu-depth', type=int, default=0, help='depth of the CPU baseline model (0 = same as --depth)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-probe', action='store_true', help='skip the MFMA probe behind roofline.sustained_peak / telemetry_probe (profiling passes: its launches would sit in the kernel stats)')
    ap.add_argument('--no-telemetry', action='store_true', help='do not sample socket power / gfx clock (AMD SMI) beside the timed region and the MFMA probe')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="infer (default, the BASELINE headline metric) | train: BASELINE config 3 - the d24 data-parallel training step, gradient "
                         "all-reduce over RCCL overlapped with the backward; reports samples/s and the exposed communication share")
    ap.add_argument('--train-batch', type=int, default=32, help='--mode train: samples per GPU per step (global 256 = 8 x 32)')
    ap.add_argument('--no-extras', action='store_true', help='skip the side configs measured after the headline region at N=1 (d12 / d30 images/s, '
                                                              'B=1 latency, VQVAE round trip, d24 training step)')
    ap.add_argument('--cpu-full', action='store_true', help='CPU baseline by the full BASELINE.md section 4 protocol (B=1 AND B=8, 1 warm-up + 3 timed '
                                                            'repetitions each, median) instead of the bounded default (B=1, short warm-up, up to 3 repetitions in ~35 s)')
    ap.add_argument('--comm-channels', default='auto', help="--mode train, N > 1: RCCL channel cap of the gradient all-reduce - 'auto' (default): try RCCL's own "
                                                            "choice, 16 and 8 for two steps each during warm-up and keep the fastest; or one value ('default', 16, 8, ...)")
    ap.add_argument('--share-gpu', action='store_true', help='--mode train, N > 1 on a box with fewer GPUs than ranks: the ranks share the visible device(s) round-robin and exchange their gradient '
                                                             'slabs over gloo (RCCL refuses two ranks on one device).  Not a scaling number - it is the real two-rank step (hand-written backward + BucketReducer + '
                                                             '1/world folding) on the only hardware a one-GPU lease offers; the line says so in config.parallelism')
    ap.add_argument('--stub-comm-ms', default='', help=argparse.SUPPRESS)                    # stub only: comma list of the fake step time of each channel candidate
    ap.add_argument('--stub-step-ms', type=float, default=0.0, help=argparse.SUPPRESS)      # tests/test_bench_launch.py: the launch / timing / JSON logic on
    return ap.parse_args(argv)                                                               # CPU ranks (gloo) with a sleeping step instead of the model


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


def cpu_baseline(depth: int, seed: int = 0, full: bool = False, budget_s: float = 35.0):
    """The oracle (kind 'port': own restatement, pinned to the reference by tests/golden) on the host cores - BASELINE.md section 4:
    fp32, greedy, cfg 4, labels arange(B) % 1000, condition types arange(B) % 4, one d{depth} generation = B samples (2B CFG rows, all ten
    scales, both VQVAE decodes), 1 warm-up + 3 timed repetitions, median, for B = 1 and B = 8.

    full=True  : exactly that (several minutes of CPU time: `bench.py --cpu-full`; the output of one such run is kept in profiles/).
    full=False : the BOUNDED default so that `python bench.py` ends within minutes - B = 1 only, the warm-up is the first six scales of
                 a generation (thread pool, allocator, oneDNN primitives), then up to 3 timed full repetitions, stopping early once
                 `budget_s` seconds are spent (at least one); median of the completed repetitions, their count is in `sample`.
    Threads: one per PHYSICAL core, and - when the host has more than 32 of them - also 32 (torch's CPU GEMMs at B = 1 do not scale to 128
    threads); the thread count with the faster warm-up runs the timed repetitions and is reported as `cores`."""
    from controlvar_amd.spec import DEFAULT_PATCH_NUMS as PN, VaeConfig, VarConfig, phi_index_map
    from controlvar_amd.synth import synth_vae_state, synth_var_state
    from oracle import var_ref
    from oracle.vqvae_ref import MSQuant
    cfg = VarConfig(depth=depth)
    sdv = synth_vae_state(VaeConfig(ch=160), seed)
    sd = synth_var_state(cfg, seed)
    msq = MSQuant(sdv, PN, phi_index_map(10))

    def generate(B, stop_after=None):
        t0 = time.time()
        hook = (lambda si: si + 1 >= stop_after) if stop_after else None
        with torch.no_grad():
            f = var_ref.generate(sd, cfg, msq, B, torch.arange(B) % 1000, 4.0, top_k=1, cond_type=torch.arange(B) % 4, stage_hook=hook)
            if not stop_after:
                var_ref.decode_fhat(sdv, f)
        return time.time() - t0

    phys = physical_cores()
    cands = [phys] + ([32] if phys > 32 else [])
    warm = {}
    for th in cands:                                        # warm-up (and thread-count choice): first six scales, B = 1
        torch.set_num_threads(th)
        warm[th] = generate(1, stop_after=6)
    threads = min(warm, key=warm.get)
    torch.set_num_threads(threads)
    res, t_start = {}, time.time()
    for B in ((1, 8) if full else (1,)):
        if full:
            generate(B)                                     # the protocol's full warm-up repetition
        reps = []
        for _ in range(3):
            reps.append(generate(B))
            if not full and time.time() - t_start > budget_s:
                break
        med = sorted(reps)[len(reps) // 2] if len(reps) != 2 else sum(reps) / 2
        res[B] = dict(img_s=B / med, median_s=med, reps=[round(r, 2) for r in reps])
    best = res[1]
    desc = '; '.join(f'B={B}: {len(r["reps"])} timed repetitions {r["reps"]} s, median {r["median_s"]:.2f} s -> {r["img_s"]:.4f} img/s' for B, r in res.items())
    out = dict(value=round(best['img_s'], 5), unit='images/s', cores=threads, kind='port',
               sample=f'd{depth} autoregressive_infer_cfg fp32 torch-CPU oracle, greedy, cfg 4, full generations incl. both VAE decodes; '
                      f'{"BASELINE.md section 4 protocol (1 warm-up + 3 timed repetitions, median)" if full else "bounded protocol (six-scale warm-up, up to 3 timed repetitions within %.0f s, median)" % budget_s}; '
                      f'{desc}; host has {phys} physical cores, warm-up per thread count: ' + ', '.join(f'{k} threads {v:.1f} s' for k, v in warm.items()))
    if 8 in res:
        out['value_b8'] = round(res[8]['img_s'], 5)
    return out


# --------------------------------------------------------------------------------------------------------------------- side configs
def _timeit(fn, steps, warmup):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def side_configs(a, dev, box):
    """The other BASELINE.json configs, measured AFTER the timed headline region on the same GPU (N = 1 only) so that the driver's bench record
    carries them: every entry names its workload and gives steps x ms_per_step (wall time the driver's clock bounds).  `box` holds the
    headline's models: the transformer is reused for the B = 1 latency and then RELEASED (its K/V arena is 154 GB), the VQVAE is kept."""
    from controlvar_amd import models, train as T
    from controlvar_amd.spec import VarConfig, algorithmic_gflop_per_row, VAE_DECODE_GFLOP, VAE_ENCODE_GFLOP
    from controlvar_amd.launcher import synthetic_rank_batch
    from controlvar_amd.synth import synth_images
    Tt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    out = {}
    var, vae = box.pop('var'), box['vae']

    def gen_rate(model, B, steps, warmup):
        labels, types = (torch.arange(B) % 1000).to(dev), (torch.arange(B) % 4).to(dev)
        dt = _timeit(lambda i: model.autoregressive_infer_cfg(B, labels, g_seed=i, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p, cond_type=types), steps, warmup)
        return dt

    # B = 1 latency of the headline model (HIP graph of the whole generation; the eager launch sequence beside it)
    l1, t1 = torch.zeros(1, dtype=torch.long, device=dev), torch.ones(1, dtype=torch.long, device=dev)
    eager = _timeit(lambda i: var.autoregressive_infer_cfg(1, l1, g_seed=i, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p, cond_type=t1), 5, 2)
    run = var.graphed_generator(1, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p)
    graph = _timeit(lambda i: run(l1, t1, g_seed=i), 10, 2)
    out['latency_b1_ms'] = {'value': round(graph * 1e3, 2), 'unit': 'ms', 'steps': 10, 'ms_per_step': round(graph * 1e3, 2), 'eager_ms': round(eager * 1e3, 2),
                            'config': f'd{a.depth} autoregressive_infer_cfg B=1 (2 CFG rows), one HIP graph per generation, incl. both decodes'}
    del run
    # the regime the reference's own sampling calls run in (train_control_var_hpu.py:372 asserts batch < 50): B = 8 and B = 32 as graphs
    for Bs in (8, 32):
        ls, ts = torch.arange(Bs, device=dev) % 1000, torch.arange(Bs, device=dev) % 4
        run = var.graphed_generator(Bs, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p)
        g = _timeit(lambda i: run(ls, ts, g_seed=i), 3, 1)
        out[f'small_batch_b{Bs}'] = {'value': round(Bs / g, 1), 'unit': 'images/s', 'steps': 3, 'ms_per_step': round(g * 1e3, 2),
                                     'config': f'd{a.depth} autoregressive_infer_cfg B={Bs}, one HIP graph per generation, incl. the decode'}
        del run
    # deterministic_plan (round 6): every transformer GEMM on the unsliced tile kernels, so that one (label, g_seed) is bit-identical at any batch size - what that costs
    # where the small-M plans matter
    try:
        var.deterministic_plan = True
        det = {}
        for Bs in (1, 8, 32):
            ls, ts = torch.arange(Bs, device=dev) % 1000, torch.arange(Bs, device=dev) % 4
            run = var.graphed_generator(Bs, cfg=a.cfg, top_k=a.top_k, top_p=a.top_p)
            g = _timeit(lambda i: run(ls, ts, g_seed=i), 3, 1)
            base = out['latency_b1_ms']['ms_per_step'] if Bs == 1 else out[f'small_batch_b{Bs}']['ms_per_step']
            det[f'b{Bs}_ms'] = round(g * 1e3, 2)
            det[f'b{Bs}_cost_vs_default_plan'] = round(g * 1e3 / base, 3)
            del run
        out['deterministic_plan'] = {'value': det['b1_ms'], 'unit': 'ms', 'steps': 3, 'ms_per_step': det['b1_ms'], **det,
                                     'config': f'd{a.depth} autoregressive_infer_cfg as one HIP graph with deterministic_plan=True (tile kernels only, no K slices: same bits at any batch size), B = 1 / 8 / 32'}
    finally:
        var.deterministic_plan = False
    var._arena = None
    del var
    torch.cuda.empty_cache()

    # BASELINE config 3 on this GPU: the d24 training step (tokenise, forward, CE, backward, clip, AdamW), B = 32
    var = models.build_control_var(vae, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=Tt).to(dev)
    Bt = a.train_batch
    tr = T.Trainer(var, vae, peak_lr=8e-5 * Bt / 512, weight_decay=0.08, sche='lin0', warmup_it=10, max_it=10000, clip=2.0, train_mode=True)
    images, masks, cls, types = synthetic_rank_batch(Bt, 0, dev)
    dt = _timeit(lambda i: tr.step(images, masks, cls, types, drop_seed=i), 4, 2)
    fl = algorithmic_gflop_per_row(VarConfig(depth=a.depth), n_ada=1)
    tf = (3 * fl['total'] + 2 * VAE_ENCODE_GFLOP) / 1e3
    out[f'train_d{a.depth}_b{Bt}'] = {'value': round(Bt / dt, 2), 'unit': 'samples/s', 'steps': 4, 'ms_per_step': round(dt * 1e3, 2), 'tflops': round(tf * Bt / dt, 1),
                                      'config': f'd{a.depth} training step, {Bt} samples, frozen tokenizer inside the step, one GPU (no exchange); {tf:.2f} TFLOP/sample'}
    del tr, var
    torch.cuda.empty_cache()

    # BASELINE config 5: VQVAE encode -> multi-scale quantise -> decode, 128 images per pass.  Round 4: ONE encode + quantise over the 128 images (the
    # ten-scale quantiser is a latency-bound 2.7 ms launch at any batch <= 256 - two chunks of 64 paid it twice); the decoder chunks by itself (decode_chunk).
    # Round 6: the three ENCODER precisions of the bf16 model side by side (the decoder is bf16 in all three), each with its id agreement MEASURED IN THIS RUN
    # against the fp32 parity mode on the same 128 images (the parity mode's ids are the reference's: 0 flips over every strict fixture,
    # tests/test_gpu_parity.py) - bf16 moves ids through its encoder's feature noise, split bf16 ("bf16x3": hi + lo operands, three MFMA products per multiply)
    # is the middle road SURVEY section 7 named, fp32 the exact-f32 MFMA for the encoder only.
    img = synth_images(128, 256, seed=3).to(dev)
    if a.dtype == 'bf16':
        vae32 = models.build_vae(ch=160, compute_dtype=torch.float32).to(dev)
        ids_ref = torch.cat(vae32.img_to_idxBl(img), dim=1)
        del vae32
        torch.cuda.empty_cache()
    for prec, key in ((('bf16', 'vqvae_roundtrip_b128'), ('bf16x3', 'vqvae_roundtrip_b128_bf16x3'), ('fp32', 'vqvae_roundtrip_b128_fp32_encoder')) if a.dtype == 'bf16'
                      else (('fp32', 'vqvae_roundtrip_b128'),)):
        v = vae if prec == vae.encoder_precision else models.build_vae(ch=160, compute_dtype=Tt, encoder_precision=prec).to(dev)
        v._pack()
        dt = _timeit(lambda i: v.idxBl_to_img(v.img_to_idxBl(img), same_shape=True, last_one=True), 3, 1)
        dte = _timeit(lambda i: v.img_to_idxBl(img), 3, 1)
        eflop = VAE_ENCODE_GFLOP * (3 if prec == 'bf16x3' else 1)
        out[key] = {'value': round(128 / dt, 1), 'unit': 'images/s', 'steps': 3, 'ms_per_step': round(dt * 1e3, 2),
                    'tflops': round((VAE_ENCODE_GFLOP + 0.23 + VAE_DECODE_GFLOP) * 128 / dt / 1e3, 1), 'encode_only_images_per_s': round(128 / dte, 1),
                    'encode_executed_tflops': round(eflop * 128 / dte / 1e3, 1), 'encoder_precision': prec,
                    'config': f'img_to_idxBl (encoder {prec}) -> idxBl_to_img(same_shape, last_one) (decoder {a.dtype}), 256^2, ch160, 128 images per pass'}
        if a.dtype == 'bf16':
            got = torch.cat(v.img_to_idxBl(img), dim=1)
            out[key]['id_agreement_vs_fp32_mode'] = round(float((got == ids_ref).float().mean()), 4)
            out[key]['id_agreement_note'] = 'measured in this run on the 128 bench images against the ids of the fp32 parity mode (= the reference\'s ids on every strict fixture)'
        if v is not vae:
            del v
            torch.cuda.empty_cache()
    del img
    if a.dtype == 'bf16':
        # the training step with EXACT-grade labels: the frozen tokenizer of the step in split bf16 (train_control_var_hpu.py:157-176 takes its labels from the fp32 encoder)
        vx3 = models.build_vae(ch=160, compute_dtype=Tt, encoder_precision='bf16x3').to(dev)
        var = models.build_control_var(vx3, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=Tt).to(dev)
        tr = T.Trainer(var, vx3, peak_lr=8e-5 * Bt / 512, weight_decay=0.08, sche='lin0', warmup_it=10, max_it=10000, clip=2.0, train_mode=True)
        images, masks, cls, types = synthetic_rank_batch(Bt, 0, dev)
        dt = _timeit(lambda i: tr.step(images, masks, cls, types, drop_seed=i), 4, 2)
        base = out[f'train_d{a.depth}_b{Bt}']
        out[f'train_d{a.depth}_b{Bt}_bf16x3_labels'] = {'value': round(Bt / dt, 2), 'unit': 'samples/s', 'steps': 4, 'ms_per_step': round(dt * 1e3, 2),
                                                        'cost_vs_bf16_labels': round(dt * 1e3 / base['ms_per_step'], 3),
                                                        'config': f'd{a.depth} training step, {Bt} samples, frozen tokenizer in split bf16 (encoder_precision=bf16x3: labels agree with the fp32 encoder\'s), one GPU'}
        del tr, var, vx3, images, masks
        torch.cuda.empty_cache()

    # the other depths of the metric's family: d12 (configs 1-2) and d30 cos-attention (config 4)
    for depth, B in ((12, 384), (30, 128)):
        if depth == a.depth:
            continue
        m = models.build_control_var(vae, depth=depth, mask_type='interleave_append', multi_cond=True, compute_dtype=Tt).to(dev).eval()
        m._pack()
        dt = gen_rate(m, B, 2, 1)
        fl = algorithmic_gflop_per_row(VarConfig(depth=depth), n_ada=1)
        tf = (2 * fl['total'] + 2 * VAE_DECODE_GFLOP) / 1e3
        out[f'd{depth}_images_per_s'] = {'value': round(B / dt, 2), 'unit': 'images/s', 'steps': 2, 'ms_per_step': round(dt * 1e3, 2), 'tflops': round(tf * B / dt, 1),
                                         'config': f'd{depth} autoregressive_infer_cfg 256^2, B={B}, same sampling settings as the headline'}
        m._arena = None
        del m
        torch.cuda.empty_cache()

    # The TOKEN-EXACT mode (VERDICT r4 missing #3): fp32 parity mode - every GEMM / conv on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain), fp32 activations -
    # is the mode whose ids are proven identical to the reference's (47 strict fixtures, 0 flips of 166 808 ids).  Its throughput, against the 157.3 TFLOP/s
    # fp32 matrix peak, so that (parity, perf) can be quoted for one and the same mode.
    if a.dtype == 'bf16':
        F32 = torch.float32
        vae32 = models.build_vae(ch=160, compute_dtype=F32).to(dev)
        vae32._pack()
        img = synth_images(32, 256, seed=3).to(dev)
        dt = _timeit(lambda i: vae32.idxBl_to_img(vae32.img_to_idxBl(img), same_shape=True, last_one=True), 2, 1)
        tf = (VAE_ENCODE_GFLOP + 0.23 + VAE_DECODE_GFLOP) * 32 / dt / 1e3
        out['fp32_vqvae_roundtrip_b32'] = {'value': round(32 / dt, 1), 'unit': 'images/s', 'steps': 2, 'ms_per_step': round(dt * 1e3, 2), 'tflops': round(tf, 1),
                                           'frac_of_fp32_matrix_peak': round(tf / 157.3, 3),
                                           'config': 'fp32 parity mode: img_to_idxBl -> idxBl_to_img(same_shape, last_one), 256^2, ch160, 32 images per pass'}
        del img
        B32 = 64
        m = models.build_control_var(vae32, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=F32).to(dev).eval()
        m._pack()
        dt = gen_rate(m, B32, 2, 1)
        fl = algorithmic_gflop_per_row(VarConfig(depth=a.depth), n_ada=1)
        tf = (2 * fl['total'] + 2 * VAE_DECODE_GFLOP) / 1e3
        out[f'fp32_d{a.depth}_b{B32}'] = {'value': round(B32 / dt, 2), 'unit': 'images/s', 'steps': 2, 'ms_per_step': round(dt * 1e3, 2), 'tflops': round(tf * B32 / dt, 1),
                                          'frac_of_fp32_matrix_peak': round(tf * B32 / dt / 157.3, 3),
                                          'config': f'fp32 parity mode (the mode whose greedy ids equal the reference CPU path token for token): d{a.depth} '
                                                    f'autoregressive_infer_cfg 256^2, B={B32}, same sampling settings as the headline, incl. both decodes'}
        m._arena = None
        del m, vae32
        torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------------- workers
def sustained_mfma(dev, launches=8, iters=6000, telemetry_s=1.5):
    """What the matrix pipes of THIS device sustain in THIS run on a register-fed stream of the product GEMM's MFMA (cvar_probe_mfma_bf16: no memory traffic, two waves
    per SIMD on every CU): on operands of the bench's kind (randn bf16) and on zeros.  MI355X clocks to its power budget, so the first is the ceiling a GEMM kernel can
    approach by scheduling alone on this box; the second shows the 2.4 GHz peak is there when nothing toggles.  Runs after the timed region: 2 x 8 launches of ~4 ms (the clock settles within the first), the median of the last four counts."""
    from controlvar_amd import _lib
    lib = _lib.load()
    res = {}
    for name, ops in (('randn', torch.randn(1 << 17, device=dev).to(torch.bfloat16)), ('zeros', torch.zeros(1 << 17, device=dev, dtype=torch.bfloat16))):
        st = torch.cuda.current_stream().cuda_stream
        sink = torch.zeros(4, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        ev[0].record()
        for i in range(launches):
            _lib.check(lib.cvar_probe_mfma_bf16(ops.data_ptr(), ops.numel() * 2, iters, sink.data_ptr(), st), 'cvar_probe_mfma_bf16')
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(launches // 2, launches))       # the later launches: the clock has settled
        res[name] = lib.cvar_probe_mfma_flops(iters) / (ms[len(ms) // 2] * 1e-3) / 1e12
        res[name + '_ghz'] = float(sink[1]) / (ms[len(ms) // 2] * 1e-3) / 1e9          # shader cycles of a wave's loop / wall time of the launch
        if telemetry_s > 0:
            # the same stream for ~telemetry_s seconds under the board sampler: socket power and gfx clock the firmware grants THIS instruction stream on THESE operands
            from controlvar_amd.telemetry import BoardSampler
            n = max(8, int(telemetry_s / (ms[len(ms) // 2] * 1e-3)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with BoardSampler(dev.index or 0) as bs:
                e0.record()
                for i in range(n):
                    _lib.check(lib.cvar_probe_mfma_bf16(ops.data_ptr(), ops.numel() * 2, iters, sink.data_ptr(), st), 'cvar_probe_mfma_bf16')
                e1.record()
                torch.cuda.synchronize()
            tel = bs.summary(skip_first_s=0.3)
            tel['tflops'] = round(n * lib.cvar_probe_mfma_flops(iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
            tel['mfma'] = 'v_mfma_f32_16x16x32_bf16'
            res[name + '_telemetry'] = tel
            # the same stream on v_mfma_f32_32x32x16_bf16 (the guide's 2 495 TFLOP/s microbenchmark shape): issue rate and clock apart
            with BoardSampler(dev.index or 0) as bs:
                e0.record()
                for i in range(n):
                    _lib.check(lib.cvar_probe_mfma_bf16_32x32(ops.data_ptr(), ops.numel() * 2, iters, sink.data_ptr(), st), 'cvar_probe_mfma_bf16_32x32')
                e1.record()
                torch.cuda.synchronize()
            tel = bs.summary(skip_first_s=0.3)
            tel['tflops'] = round(n * lib.cvar_probe_mfma_flops(iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
            tel['mfma'] = 'v_mfma_f32_32x32x16_bf16'
            ticks, real = float(sink[1]), float(sink[2])
            tel['s_memtime_ticks_per_us'] = round(ticks / (real / 100.0), 1) if real > 0 else None      # s_memrealtime runs at 100 MHz
            res[name + '_telemetry_32x32'] = tel
    return res


def _board_keys(board, probe):
    """roofline.power_w / power_cap_w / sclk_mhz (+ the same for the MFMA probe): the board's own account of the timed region (controlvar_amd/telemetry.py)"""
    if board is None:
        return {}
    t = board.summary(skip_first_s=0.3)
    out = {'power_w': t.get('power_w'), 'power_cap_w': t.get('power_cap_w'), 'sclk_mhz': t.get('sclk_mhz'), 'telemetry': t}
    if probe:
        out['telemetry_probe'] = probe
        out['telemetry_note'] = ('AMD SMI gpu_metrics sampled every 20 ms by a thread of this process over the timed region (telemetry) and over ~1.5 s of the MFMA probe on randn / zero '
                                 'operands (telemetry_probe): socket power against the cap, mean gfx clock over the XCDs, and the firmware power-limit residency over the region')
    return out


def _finish(world):
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def _rank_evidence(units_per_step, steps, device):
    """keys that let the one JSON line prove the N-rank run by itself: rccl_ranks (from a real all_reduce), rccl_version, per-rank rates"""
    from controlvar_amd.launcher import LAST_RUN, collective_evidence
    ev = collective_evidence(device)
    ev['per_rank_value'] = [round(units_per_step * steps / s, 3) for s in LAST_RUN['rank_seconds']]
    return ev


def _channel_candidates(a):
    if a.comm_channels == 'auto':
        return [None, 16, 8]
    return [None if a.comm_channels in ('default', '0', 'none') else int(a.comm_channels)]


def _tune_channels(a, world, device, run_two_steps):
    """the first 8-GPU contact tunes itself.  For every candidate channel cap (launcher.channel_groups: a process group
    whose communicator is capped at that many channels = workgroups = CUs taken from the backward GEMMs) run `run_two_steps(label, group)`
    - an untimed pair `run_two_steps(label, group, True)` that selects the group and builds the communicator, then a timed pair (`..., False`) -,
    agree on the max over ranks and keep the fastest.
    Returns the keys for the JSON line; the chosen group is in the returned dict under '_group' (popped by the caller)."""
    from controlvar_amd.launcher import channel_groups, pick_fastest
    cands = _channel_candidates(a)
    import torch.distributed as dist
    err = None
    try:
        groups = channel_groups(cands)
    except Exception as e:                   # a library that rejects the per-communicator cap must not cost the run its line: default group, reason on the line
        groups, err = None, f'{type(e).__name__}: {e}'[:300]
    # every rank must take the same branch: a cap that failed on ONE rank would leave the others in pick_fastest's collectives
    ok = torch.tensor([0.0 if groups is None else 1.0], device=device if (device is not None and dist.is_initialized() and dist.get_backend() == 'nccl') else 'cpu')
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 1.0:
        return {'comm_channels': 'default', 'comm_channels_tried_s': {}, 'comm_channels_error': err or 'channel_groups failed on another rank', '_group': None}

    def seconds_of(label):
        run_two_steps(label, groups[label], True)               # selects the group (reducer construction) + communicator first use: outside the clock
        if device is not None:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        run_two_steps(label, groups[label], False)
        if device is not None:
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    if len(cands) == 1:
        best, table = cands[0], {}
    else:
        best, table = pick_fastest(cands, seconds_of, device)
    name = lambda c: 'default' if c is None else str(c)
    return {'comm_channels': name(best), 'comm_channels_tried_s': {name(k): round(v, 4) for k, v in table.items()}, '_group': groups[best]}


def main_stub(a):
    """the launch / barrier / max-over-ranks clock / one-JSON-line logic with a sleeping step on CPU ranks (gloo): what
    tests/test_bench_launch.py runs with --gpus 2 to check that `python bench.py --gpus N` really becomes N ranks"""
    from controlvar_amd.launcher import dist_env, init_dist, sharded_timed_run
    rank, local, world = dist_env()
    init_dist('gloo')
    B = a.batch or 4
    tuned = {}
    if a.stub_comm_ms and world > 1:
        # the channel selection of main_train with sleeping candidates: rank r sleeps (1 + r) x the candidate's milliseconds, so the agreed
        # time of a candidate is the SLOWEST rank's and every rank must land on the same choice
        fake = dict(zip(_channel_candidates(a), [float(x) for x in a.stub_comm_ms.split(',')]))
        tuned = _tune_channels(a, world, None, lambda label, group, select: time.sleep(fake[label] * 1e-3 * (1 + rank)))
    _, dt = sharded_timed_run(lambda i: time.sleep(a.stub_step_ms * 1e-3 * (1 + rank)), a.steps, a.warmup, B)
    tuned.pop('_group', None)
    ev = {**_rank_evidence(B, a.steps, None), **tuned}
    if rank == 0:
        print(json.dumps({**ev, 'metric': 'stub', 'value': round(world * B * a.steps / dt, 3), 'unit': 'units/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                          'ms_per_step': round(1e3 * dt / a.steps, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none',
                          'data': 'stub', 'config': {'workload': 'sleeping step (launch-logic test)', 'batch_per_gpu': B, 'global_batch': B * world,
                                                     'parallelism': f'dp{world}'}}), flush=True)
    _finish(world)


def main_train(a):
    """BASELINE config 3: d24 joint image+control training, synthetic ImageNetC-shaped batch of --train-batch samples per GPU (seeded by
    the rank), frozen tokenizer inside the step, per-layer gradient slabs all-reduced (SUM, mean folded into AdamW) over RCCL on a side
    stream while the backward continues.  The same K steps are timed once more with the exchange switched off; the difference is the
    EXPOSED communication time (what the overlap did not hide)."""
    from controlvar_amd.launcher import dist_env, init_dist, sharded_timed_run, synthetic_rank_batch
    rank, local, world = dist_env()
    if a.share_gpu:
        local = local % max(1, torch.cuda.device_count())
        if a.comm_channels == 'auto':
            a.comm_channels = 'default'                         # channel caps are an RCCL setting
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    init_dist('gloo' if a.share_gpu else 'nccl', None if a.share_gpu else dev)
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

    tuned = {}
    if world > 1:
        def two_steps(label, group, select):
            if select:                          # once per candidate, before the untimed pair: the reducer is rebuilt outside the clock
                tr.set_comm_group(group)
            step(0); step(1)
        tuned = _tune_channels(a, world, dev, two_steps)
        tr.set_comm_group(tuned.pop('_group'))
    _, dt = sharded_timed_run(step, a.steps, a.warmup, B, sync=torch.cuda.synchronize)
    ev = {**_rank_evidence(B, a.steps, dev), **tuned}
    if ev['rccl_ranks'] != world:
        sys.exit(f'[bench] the collective saw {ev["rccl_ranks"]} rank(s), the job has {world}: refusing to print a line for a job that is not the one asked for')
    exposed, allreduce_ms = None, None
    if world > 1:
        tr.comm = False
        _, dt_nocomm = sharded_timed_run(step, a.steps, 1, B, sync=torch.cuda.synchronize)
        tr.comm = True
        exposed = max(0.0, (dt - dt_nocomm) / dt)
        # the exchange alone, un-overlapped: every gradient slab of one step all-reduced back to back (what the overlap has to hide)
        import torch.distributed as dist
        slabs = [torch.zeros_like(b) for b in tr.engine.buckets]
        for rep in range(2):
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for s_ in slabs:
                dist.all_reduce(s_, op=dist.ReduceOp.SUM, group=tr.comm_group)
            torch.cuda.synchronize()
            allreduce_ms = 1e3 * (time.perf_counter() - t0)
        del slabs
    if rank == 0:
        fl = algorithmic_gflop_per_row(VarConfig(depth=a.depth), n_ada=1)
        per_sample_tf = (3 * fl['total'] + 2 * 215.4) / 1e3                   # fwd + 2x bwd + two frozen tokenizer encodes
        out = {'metric': 'training samples/sec (d%d joint image+control, DP all-reduce)' % a.depth, 'value': round(world * B * a.steps / dt, 2),
               'unit': 'samples/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': f'ControlVAR d{a.depth} training step (tokenise image+control, forward, CE, backward, clip 2.0, AdamW), '
                                      f'synthetic ImageNetC-shaped batch', 'batch_per_gpu': B, 'global_batch': B * world, 'seq_len': 1360,
                          'parallelism': (f'dp{world} SHARING {torch.cuda.device_count()} GPU(s), per-layer gradient slabs all-reduced over gloo on a side stream - the real two-rank step, not a scaling number'
                                          if a.share_gpu else f'dp{world} (per-layer gradient slabs, RCCL all-reduce on a side stream)')},
               'algorithmic_tflop_per_sample': round(per_sample_tf, 3), 'end_to_end_tflops_per_gpu': round(per_sample_tf * B * a.steps / dt, 1),
               'loss': round(float(last['out']['loss']), 4), 'exposed_comm_frac': None if exposed is None else round(exposed, 4),
               'allreduce_ms_per_step': None if allreduce_ms is None else round(allreduce_ms, 2), **ev,
               'allreduce_bytes_per_step': sum(b.numel() * 4 for b in tr.engine.buckets)}
        print(json.dumps(out), flush=True)
    _finish(world)


def main_infer(a):
    from controlvar_amd.launcher import dist_env, init_dist, sharded_timed_run
    rank, local, world = dist_env()
    torch.cuda.set_device(local)
    init_dist('nccl', torch.device('cuda', local))
    if a.gpus != world and rank == 0:
        print(f'[bench] --gpus {a.gpus} != WORLD_SIZE {world}: the environment wins, running {world} rank(s)', file=sys.stderr)
    dev = torch.device('cuda', local)

    from controlvar_amd import models, ops, _lib
    from controlvar_amd.spec import VarConfig, algorithmic_gflop_per_row, VAE_DECODE_GFLOP
    _lib.load()
    T = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    t_build = time.time()
    vae = models.build_vae(ch=160, compute_dtype=T).to(dev)
    if a.decode_chunk > 0:
        vae.decode_chunk = a.decode_chunk
    var = models.build_control_var(vae, depth=a.depth, mask_type='interleave_append', multi_cond=True, compute_dtype=T).to(dev).eval()
    B = a.batch or (512 if a.depth <= 24 else 128)       # round 4: 384 -> 512 (+1.0 % on one box: 195.4 -> 197.3 images/s; 576: 198.1 at 254 GB - not taken)
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

    board = None
    if rank == 0 and not a.no_telemetry:
        from controlvar_amd.telemetry import BoardSampler
        board = BoardSampler(local)

    def timed_step(i):
        if i == a.warmup:
            ops.GEMM_PROFILE = prof                     # kernel events only inside the timed region
            if board is not None:
                board.start()                           # socket power / gfx clock of the timed region (a thread polling AMD SMI every 20 ms)
        last['img'] = step(100 + i)

    _, dt = sharded_timed_run(timed_step, a.steps, a.warmup, B, sync=torch.cuda.synchronize)
    if board is not None:
        board.stop()
    ops.GEMM_PROFILE = None
    ev = _rank_evidence(B, a.steps, dev)
    if ev['rccl_ranks'] != world:
        sys.exit(f'[bench] the collective saw {ev["rccl_ranks"]} rank(s), the job has {world} (--gpus {a.gpus}): refusing to print a line for another job')
    img = last.pop('img')
    assert img.shape == (B, 3, 512, 256)
    del img

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
            'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), **ev,
        }
        if prof:
            ms = sum(r[0].elapsed_time(r[1]) for r in prof)
            flops = sum(r[2] for r in prof)
            ach = flops / (ms * 1e-3) / 1e12
            peak = 2500.0 if a.dtype == 'bf16' else 157.3
            sustained = {}
            telemetry_probe = None
            if a.dtype == 'bf16' and not a.no_probe:
                try:
                    sm = sustained_mfma(dev, telemetry_s=0.0 if a.no_telemetry else 1.5)
                    if 'randn_telemetry' in sm:
                        telemetry_probe = {'randn_operands': sm['randn_telemetry'], 'zero_operands': sm['zeros_telemetry'],
                                           'randn_operands_32x32x16': sm.get('randn_telemetry_32x32'), 'zero_operands_32x32x16': sm.get('zeros_telemetry_32x32')}
                    sustained = {'sustained_peak': round(sm['randn'], 1), 'frac_of_sustained': round(ach / sm['randn'], 4), 'peak_on_zero_operands': round(sm['zeros'], 1),
                                 'clock_ghz_randn_zeros': [round(sm['randn_ghz'], 2), round(sm['zeros_ghz'], 2)],
                                 'sustained_note': 'cvar_probe_mfma_bf16 in this run on this device: a register-fed stream of the GEMM\'s MFMA (v_mfma_f32_16x16x32_bf16, two waves '
                                                   'per SIMD on every CU, no memory traffic) on randn bf16 operands / on zeros - the part clocks to its power budget, so '
                                                   'sustained_peak is the ceiling a GEMM kernel can approach by scheduling alone; frac stays against the 2.4 GHz peak'}
                except Exception as e:
                    sustained = {'sustained_peak': None, 'sustained_note': f'probe failed: {type(e).__name__}: {e}'}
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, 'profiles', 'gemm_hbm_traffic.json')
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get('bytes_per_launch')
                    # the counter passes are separate rocprofv3 runs (gpurun refuses --pmc beside tracing): the file is stamped with the digest of the kernel
                    # sources it was measured on (controlvar_amd/csrc/build/digest.txt; .git does not travel to the GPU box) - say whether that is THIS library
                    try:
                        here = open(os.path.join(ROOT, 'controlvar_amd', 'csrc', 'build', 'digest.txt')).read().strip()[:16]
                    except OSError:
                        here = None
                    was = tj.get('lib_digest')
                    same = 'unknown (file carries no stamp)' if not was else ('this library' if was == here else f'ANOTHER library build ({was}; this one: {here})')
                    tsrc = f'profiles/gemm_hbm_traffic.json ({tj.get("collected", "separate rocprofv3 --pmc passes of this command")}); measured on: {same}; not re-measured in this run'
                except Exception:
                    traffic = None
            out['roofline'] = {'bound': 'mfma', 'kernel': 'cvar_gemm_kernel + conv3x3_halo_bf16_kernel (every cvar_gemm launch: GEMMs and 3x3 convs)',
                               'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                               'traffic': traffic, 'traffic_source': tsrc, 'launches': len(prof), 'avg_launch_ms': round(ms / len(prof), 4),
                               'gemm_share_of_step': round(ms * 1e-3 / dt, 3), **sustained, **_board_keys(board, telemetry_probe),
                               'peak_note': 'dense bf16 MFMA peak at 2.4 GHz; this kernel is power-limited on random operands (the same instruction stream runs '
                                            '~30 % faster on constant operands: profiles/r03_gemm_power.txt), so the clock under load is ~1.85 GHz'}
        prof = None
        if world == 1 and not a.no_extras:
            try:
                box = {'var': var, 'vae': vae}
                var = None
                out['side_configs'] = side_configs(a, dev, box)
            except Exception as e:                                                  # a side config must never lose the headline line
                out['side_configs'] = {'error': f'{type(e).__name__}: {e}'}
        if not a.no_cpu_baseline and world == 1:            # reported baseline: rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(a.cpu_depth or a.depth, full=a.cpu_full)
        print(json.dumps(out), flush=True)
    _finish(world)


def run(a):
    if a.stub_step_ms > 0:
        return main_stub(a)
    return main_train(a) if a.mode == 'train' else main_infer(a)


def _spawned(rank, world, argv):
    run(parse(argv))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks here (the reference: mp.spawn(main_worker, nprocs=ngpus_per_node),
        # train_control_var_hpu.py:692-697).  Under torchrun WORLD_SIZE is set and this process is already one of the ranks.
        stub = a.stub_step_ms > 0
        if not stub:
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < a.gpus and not (a.share_gpu and a.mode == 'train' and have >= 1):
                sys.exit(f'[bench] --gpus {a.gpus} but only {have} GPU(s) are visible to this process - refusing to run a smaller job under the same name')
        from controlvar_amd.launcher import spawn
        spawn(_spawned, nprocs=a.gpus, args=(argv,), backend='gloo' if (stub or a.share_gpu) else 'nccl')
        return
    run(a)


if __name__ == '__main__':
    main()
