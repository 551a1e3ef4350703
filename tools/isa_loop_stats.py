#!/usr/bin/env python3
"""Instruction mix of a GEMM main loop, per loop iteration and wave: counts by class, the longest run of MFMAs with nothing between them, what
stands between the loop head and the first MFMA.  Input: a text file holding ONE loop body (llvm-objdump or hipcc -S text), e.g.

  hipBLASLt (yardstick):  clang-offload-bundler --unbundle ... TensileLibrary_BB_BB_HA_Bias_SAV_UA_Type_BB_HPA_Contraction_l_Alik_Bljk_Cijk_Dijk_gfx950.co,
                          llvm-objdump -d, label_LoopBeginL0 .. s_cbranch label_LoopBeginL0
  this repo:              hipcc --offload-arch=gfx950 ... --cuda-device-only -S gemm.hip, the '=>This Inner Loop Header' block of the K loop

usage: isa_loop_stats.py name=file [name=file ...]"""
import re, sys

CLASSES = [('mfma', r'^v_mfma'), ('ds_read', r'^ds_read'), ('ds_write', r'^ds_write'), ('dma (buffer_load ... lds)', r'^buffer_load.*\blds\b'),
           ('buffer/global load (to VGPR)', r'^(buffer|global)_load(?!.*\blds\b)'), ('s_waitcnt', r'^s_waitcnt'), ('s_barrier', r'^s_barrier'),
           ('valu (non-mfma)', r'^v_(?!mfma)'), ('salu / m0', r'^s_(?!waitcnt|barrier|cbranch|branch|nop)'), ('branch', r'^s_c?branch'), ('s_nop', r'^s_nop')]


def stats(path):
    ins = []
    for line in open(path):
        line = line.split('//')[0].split(';')[0].strip()
        if not line or line.endswith(':') or line.startswith('.') or re.match(r'^[0-9a-f]+ <', line):
            continue
        ins.append(line)
    out = {k: sum(1 for i in ins if re.match(r, i)) for k, r in CLASSES}
    out['instructions'] = len(ins)
    mf = [re.match(r'^v_mfma', i) is not None for i in ins]
    run = best = 0
    for m in mf:
        run = run + 1 if m else 0
        best = max(best, run)
    out['longest back-to-back mfma run'] = best
    first = mf.index(True) if True in mf else 0
    head = ins[:first]
    out['before the first mfma'] = ', '.join(f'{sum(1 for i in head if re.match(r, i))} {k}' for k, r in CLASSES if any(re.match(r, i) for i in head)) or 'nothing'
    out['waitcnt forms'] = ', '.join(sorted({i.replace('s_waitcnt ', '') for i in ins if i.startswith('s_waitcnt')}))
    shapes = sorted({re.match(r'^(v_mfma\w+)', i).group(1) for i in ins if i.startswith('v_mfma')})
    out['mfma shape'] = ', '.join(shapes)
    widths = sorted({re.match(r'^(ds_read\w+)', i).group(1) for i in ins if i.startswith('ds_read')})
    out['ds_read width'] = ', '.join(widths)
    return out


if __name__ == '__main__':
    cols = [a.split('=', 1) for a in sys.argv[1:]]
    res = [(n, stats(p)) for n, p in cols]
    keys = list(res[0][1])
    w = max(len(k) for k in keys) + 2
    print(' ' * w + ''.join(f'{n:>28s}' for n, _ in res))
    for k in keys:
        vals = [str(r[k]) for _, r in res]
        if max(len(v) for v in vals) > 26:
            print(f'{k}:')
            for (n, _), v in zip(res, vals):
                print(f'    {n}: {v}')
        else:
            print(f'{k:<{w}s}' + ''.join(f'{v:>28s}' for v in vals))
