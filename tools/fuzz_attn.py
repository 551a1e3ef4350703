#!/usr/bin/env python3
"""Randomised sweep of the MFMA flash attention against the exact row-wise kernel (same library) and a torch fp32 reference:
random (rows, heads, query span, cache length, level masks), the qkv arena embedded in a NaN-filled buffer so that any read
outside the visible keys / the arena shows up.  usage: fuzz_attn.py [n_cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlvar_amd import ops

dev = torch.device('cuda:0'); T = torch.bfloat16
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = [int(x) for x in os.environ.get('FUZZ_ONLY', '').split(',') if x]
bad = 0


def visible(Lmax, levels, holes):
    """(Lmax, Lmax) bool: query p sees keys [0, end(level(p))) minus its level's hole - the contract of cvar_attention"""
    vis = torch.zeros(Lmax, Lmax, dtype=torch.bool)
    b = 0
    for k, e in enumerate(levels):
        vis[b:e, :e] = True
        if holes and holes[k][1] > holes[k][0]:
            vis[b:e, holes[k][0]:holes[k][1]] = False
        b = e
    return vis


def torch_ref(qkv, R, H, Lmax, q_off, l, scale, levels, holes=None):
    """fp32 softmax(q k^T * scale + mask) v over the visible keys"""
    q, k, v = qkv.float().view(R, Lmax, 3, H, 64).permute(2, 0, 3, 1, 4).unbind(0)
    q = q[:, :, q_off:q_off + l]
    nk = q_off + l
    s = torch.matmul(q, k[:, :, :nk].transpose(-1, -2)) * scale
    if levels is not None:
        s = s.masked_fill(~visible(Lmax, levels, holes)[q_off:q_off + l, :nk].to(qkv.device), float('-inf'))
    return torch.matmul(torch.softmax(s, -1), v[:, :, :nk]).transpose(1, 2).reshape(R * l, H * 64)


for case in range(n_cases):
    R, H = rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 4, 12])
    levels, holes = None, None
    if rng.random() < 0.4:                                    # inference: queries [q_off, q_off + l) see keys [0, q_off + l)
        Lmax = rng.choice([40, 130, 300, 700, 1360])
        l = rng.randint(1, min(Lmax, 520))
        q_off = rng.randint(0, Lmax - l)
    else:                                                     # training: block-causal level mask over the whole sequence
        pns = rng.choice([(1, 2, 3), (1, 2, 3, 4, 5, 6), (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)])
        ends, acc = [], 0
        for p in pns:
            acc += 2 * p * p; ends.append(acc)
        Lmax, l, q_off, levels = acc, acc, 0, ends
        kind = rng.random()
        if kind < 0.6:                                        # separate_decoding: half-scale levels (+ indep holes), random sub-span of queries
            lv, hl, b0 = [], [], 0
            for e in ends:
                half = (e - b0) // 2
                lv += [b0 + half, e]
                hl += [(0, 0), (b0, b0 + half) if kind < 0.35 else (0, 0)]
                b0 = e
            levels, holes = lv, (hl if kind < 0.35 else None)
            if rng.random() < 0.5:                            # KV-cached form with the mask rows (indep inference): one scale's queries
                k = rng.randrange(len(ends))
                q_off = 0 if k == 0 else ends[k - 1]
                l = ends[k] - q_off
    C3 = 3 * H * 64
    g = torch.Generator().manual_seed(case)
    amp = rng.choice([0.3, 1.0, 2.5])
    qkv_cpu = (torch.randn(R, Lmax, C3, generator=g) * amp).to(T)
    pad = 8192
    buf = torch.full((qkv_cpu.numel() + 2 * pad,), float('nan'), device=dev, dtype=T)
    qkv = buf[pad:pad + qkv_cpu.numel()].view(R, Lmax, C3)
    qkv.copy_(qkv_cpu)
    # keys the queries must not see are poisoned as well (inference: rows >= q_off + l)
    if q_off + l < Lmax:
        qkv[:, q_off + l:, H * 64:] = float('nan')
    scale = rng.choice([0.125, 0.03125, 1.0])
    if only and case not in only:
        continue
    out = torch.empty(R * l, H * 64, device=dev, dtype=T)
    ref = torch.empty(R * l, H * 64, device=dev, dtype=T)
    ops.attention(qkv, out, R, H, Lmax, q_off, l, scale, levels, holes=holes)
    ops.attention(qkv, ref, R, H, Lmax, q_off, l, scale, levels, rowwise=True, holes=holes)
    # K/V-arena form (queries in their own NaN-fenced buffer): must be bit-identical to the packed form
    Cq = H * 64
    qbuf = torch.full((R * l * Cq + 2 * pad,), float('nan'), device=dev, dtype=T)
    qs = qbuf[pad:pad + R * l * Cq].view(R, l, Cq)
    qs.copy_(qkv[:, q_off:q_off + l, :Cq])
    kvbuf = torch.full((R * Lmax * 2 * Cq + 2 * pad,), float('nan'), device=dev, dtype=T)
    kv = kvbuf[pad:pad + R * Lmax * 2 * Cq].view(R, Lmax, 2 * Cq)
    kv.copy_(qkv[:, :, Cq:])
    out2 = torch.empty_like(out)
    ops.attention(kv, out2, R, H, Lmax, q_off, l, scale, levels, holes=holes, q=qs)
    if not torch.equal(out2, out):
        bad += 1
        print('FAIL', case, 'K/V-arena form differs from the packed form', (out2.float() - out.float()).abs().max().item(), flush=True)
        continue
    # the inference path proper (round 3): queries pre-multiplied by scale * log2 e, maximum subtracted by the bias k-step.  Checked against the
    # exact row-wise kernel run on the SAME prescaled (re-rounded) queries with scale = ln 2, i.e. the identical function of identical operands.
    qps_buf = torch.full((R * l * Cq + 2 * pad,), float('nan'), device=dev, dtype=T)
    qps = qps_buf[pad:pad + R * l * Cq].view(R, l, Cq)
    qps.copy_((qs.float() * (scale * 1.4426950408889634)).to(T))
    out3, ref3 = torch.empty_like(out), torch.empty_like(out)
    ops.attention(kv, out3, R, H, Lmax, q_off, l, scale, levels, holes=holes, q=qps, prescaled=True)
    ops.attention(kv, ref3, R, H, Lmax, q_off, l, 0.6931471805599453, levels, holes=holes, q=qps, rowwise=True)
    a3, b3 = out3.float(), ref3.float()
    err3 = ((a3 - b3).abs() / (b3.abs() + 0.05 * max(1.0, amp))).max().item() if torch.isfinite(a3).all() and torch.isfinite(b3).all() else float('nan')
    if not (err3 < 0.12):
        # two kernels that both round P to bf16 can differ by the SUM of their errors on near-one-hot rows (|logit| ~ 50-100 at scale 1, amplitude
        # 2.5): the judge is the exact softmax (float64) of the same operands - each kernel alone must stay inside the bound
        kz = kv.clone(); kz[~torch.isfinite(kz)] = 0
        kk = kz[:, :, :Cq].double().view(R, Lmax, H, 64).permute(0, 2, 1, 3)
        vv = kz[:, :, Cq:].double().view(R, Lmax, H, 64).permute(0, 2, 1, 3)
        qq = qps.double().view(R, l, H, 64).permute(0, 2, 1, 3)
        nk = q_off + l
        sc = torch.matmul(qq, kk[:, :, :nk].transpose(-1, -2)) * 0.6931471805599453
        if levels is not None:
            sc = sc.masked_fill(~visible(Lmax, levels, holes)[q_off:q_off + l, :nk].to(dev), float('-inf'))
        ex = torch.matmul(torch.softmax(sc, -1), vv[:, :, :nk]).transpose(1, 2).reshape(R * l, Cq).float()
        e_m = ((a3 - ex).abs() / (ex.abs() + 0.05 * max(1.0, amp))).max().item() if torch.isfinite(a3).all() else float('nan')
        e_r = ((b3 - ex).abs() / (ex.abs() + 0.05 * max(1.0, amp))).max().item()
        if not (e_m < 0.15):            # measured over 1 500 cases (seeds 11-13): worst 0.123 at scale 1.0 (|logit| ~ 100: one bf16 ulp of a near-zero output); the row-wise kernel's worst 0.09
            bad += 1
            print('FAIL', case, 'prescaled kernel vs exact softmax of the same operands', dict(R=R, H=H, Lmax=Lmax, l=l, q_off=q_off, scale=scale, amp=amp),
                  'mfma', e_m, 'row-wise', e_r, 'mfma vs row-wise', err3, flush=True)
            continue
    if not (torch.isfinite(qps_buf[:pad]).sum() == 0 and torch.isfinite(qps_buf[pad + R * l * Cq:]).sum() == 0):
        bad += 1
        print('FAIL', case, 'fence around the prescaled queries damaged', flush=True)
        continue
    a, b = out.float(), ref.float()
    err = ((a - b).abs() / (b.abs() + 0.05 * max(1.0, amp))).max().item() if torch.isfinite(a).all() and torch.isfinite(b).all() else float('nan')
    ok = err == err and err < 0.12          # bf16 P vs exact fp32 softmax; near one-hot rows (scale 1.0, large logits) sit at 0.07-0.09
    # (the floor scales with the value amplitude: at amp 2.5 a one-ulp bf16 flip of an output near 4-8 is 0.031 absolute)
    if ok and levels is not None and Lmax <= 400:             # common-mode check of both kernels against torch on the masked structures
        vis0 = qkv.clone(); vis0[~torch.isfinite(vis0)] = 0
        tt = torch_ref(vis0, R, H, Lmax, q_off, l, scale, levels, holes)
        e2 = ((b - tt).abs() / (tt.abs() + 0.05 * max(1.0, amp))).max().item()
        if not (e2 < 0.12):             # the row-wise kernel rounds P to bf16 like the MFMA kernel: same rounding-level floor as above
            ok, err = False, e2
    if not ok:
        bad += 1
        vis = qkv.clone()
        vis[~torch.isfinite(vis)] = 0
        t = torch_ref(vis, R, H, Lmax, q_off, l, scale, levels, holes)
        print('FAIL', case, dict(R=R, H=H, Lmax=Lmax, l=l, q_off=q_off, levels=len(levels) if levels else 0, holes=bool(holes), scale=scale, amp=amp), 'err', err,
              '| max abs mfma-rowwise', (a - b).abs().max().item(), 'mfma-fp32', (a - t).abs().max().item(), 'rowwise-fp32', (b - t).abs().max().item(),
              'mean abs mfma-fp32', (a - t).abs().mean().item(), 'rowwise-fp32', (b - t).abs().mean().item(), flush=True)
print(f'{n_cases - bad}/{n_cases} cases ok')
sys.exit(1 if bad else 0)
