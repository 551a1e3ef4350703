#!/usr/bin/env python3
"""Per-scale / per-kernel-family time of ONE generation out of a rocprofv3 --kernel-trace CSV of tools/b1_profile.py.
The generations are cut at first_tokens_kernel, the scales at cfg_sample_kernel (one per scale); what follows the tenth sampler is the decode.
usage: trace_by_scale.py <kernel_trace.csv> [generation index, default last]"""
import csv, sys, collections

FAM = [('gemm128', 'cvar_gemm_kernel<unsigned short, 128, 128'), ('gemm64', 'cvar_gemm_kernel<unsigned short, 64, 128'),
       ('gemm256', 'cvar_gemm_kernel<unsigned short, 256, 256'), ('gemm192', 'cvar_gemm_kernel<unsigned short, 256, 192'), ('gemmskinny', 'skinny'), ('splitk_epi', 'cvar_splitk_epilogue'), ('rowfin', 'rowfin'),
       ('attn', 'attn_'), ('ln_mod', 'ln_modulate'), ('conv_halo', 'conv3x3_halo'), ('gemm_conv', 'cvar_gemm_kernel'), ('gn', 'gn_'),
       ('sampler', 'cfg_sample'), ('msq', 'ms_'), ('word_embed', 'word_embed')]


def fam(name):
    for k, pat in FAM:
        if pat in name:
            return k
    return 'other'


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    gens, cur = [], None
    for r in rows:
        if 'first_tokens_kernel' in r['Kernel_Name']:
            cur = []
            gens.append(cur)
        if cur is not None:
            cur.append(r)
    g = gens[int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else -1]
    t0 = int(g[0]['Start_Timestamp'])
    scale = 0
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    span = collections.defaultdict(lambda: [None, None])
    for r in g:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = fam(r['Kernel_Name'])
        sc = min(scale, 10)
        per[sc][k][0] += 1
        per[sc][k][1] += (e - s) / 1e3
        if span[sc][0] is None:
            span[sc][0] = s
        span[sc][1] = e
        if 'cfg_sample' in r['Kernel_Name']:
            scale += 1
    fams = sorted({k for sc in per for k in per[sc]}, key=lambda k: -sum(per[sc][k][1] for sc in per if k in per[sc]))
    print('generation: %d launches, span %.2f ms (profiled)' % (len(g), (int(g[-1]['End_Timestamp']) - t0) / 1e6))
    print('%-8s %9s %9s  ' % ('scale', 'span_us', 'busy_us') + ' '.join('%16s' % f for f in fams))
    tot = collections.defaultdict(lambda: [0, 0.0])
    for sc in sorted(per):
        busy = sum(v[1] for v in per[sc].values())
        cells = []
        for f in fams:
            n, t = per[sc].get(f, [0, 0.0])
            tot[f][0] += n; tot[f][1] += t
            cells.append('%5d x%9.1f' % (n, t) if n else ' ' * 16)
        print('%-8s %9.1f %9.1f  ' % ('decode' if sc == 10 else sc, (span[sc][1] - span[sc][0]) / 1e3, busy) + ' '.join(cells))
    print('%-8s %9s %9.1f  ' % ('total', '', sum(v[1] for v in tot.values())) + ' '.join('%5d x%9.1f' % (tot[f][0], tot[f][1]) for f in fams))
    if '--decode' in sys.argv:
        # the launches behind the tenth sampler, in order: family, grid (workgroups), duration
        n = 0
        print('decode launches (family, workgroups x threads, us):')
        for r in g:
            if n >= 10:
                wg = int(r.get('Workgroup_Size_X') or 0) * max(int(r.get('Workgroup_Size_Y') or 1), 1) * max(int(r.get('Workgroup_Size_Z') or 1), 1)
                gs = int(r.get('Grid_Size_X') or 0) * max(int(r.get('Grid_Size_Y') or 1), 1) * max(int(r.get('Grid_Size_Z') or 1), 1)
                print('  %-10s %6d x %4d  %8.1f  %s' % (fam(r['Kernel_Name']), gs // max(wg, 1), wg, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:60]))
            if 'cfg_sample' in r['Kernel_Name']:
                n += 1


if __name__ == '__main__':
    main()
