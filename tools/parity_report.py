#!/usr/bin/env python3
"""Integer / bf16 parity report (VERDICT r2, next #1c).

    python tools/parity_report.py [--run] [--out profiles/r04_parity_report.json]

--run   runs `pytest tests -m gpu` on this box with a fresh JSON-lines record file (tests/conftest.py: `record`, `ids_parity`), then
        summarises; without it the existing gpurun_out/parity_report.jsonl is summarised.
Output: per fixture the number of ids compared, the number that differ from the reference-recorded ids (flips), whether the assertion was
strict (zero flips) and the largest reference margin at a flip; per bf16 comparison the four logit distances (max, RMS relative to
max|logit|): HIP-bf16 vs reference fp32, reference bf16-autocast vs reference fp32, HIP-bf16 vs reference autocast, HIP-bf16 vs emulation."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--run', action='store_true')
    ap.add_argument('--jsonl', default=os.path.join(ROOT, 'gpurun_out', 'parity_report.jsonl'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r04_parity_report.json'))
    a = ap.parse_args()
    rc = None
    if a.run:
        if os.path.exists(a.jsonl):
            os.remove(a.jsonl)
        env = dict(os.environ, CVAR_PARITY_REPORT=a.jsonl)
        rc = subprocess.call([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests'), '-q', '-m', 'gpu', '-x'], env=env, cwd=ROOT)
    recs = [json.loads(l) for l in open(a.jsonl)] if os.path.exists(a.jsonl) else []
    ids, bf, other = {}, {}, {}
    for r in recs:
        kind = r.get('kind')
        (ids if kind == 'ids' else (bf if kind in ('bf16_logits', 'bf16', None) else other))[r['what']] = {k: v for k, v in r.items() if k != 'what'}     # last run of a name wins
    strict = {k: v for k, v in ids.items() if v.get('strict')}
    out = {
        'pytest_rc': rc,
        'summary': {'fixtures_strict': len(strict), 'ids_compared_strict': sum(v['total'] for v in strict.values()),
                    'flips_strict': sum(v['flips'] for v in strict.values()),
                    'fixtures_margin_bounded': len(ids) - len(strict), 'ids_compared_margin_bounded': sum(v['total'] for k, v in ids.items() if k not in strict),
                    'flips_margin_bounded': sum(v['flips'] for k, v in ids.items() if k not in strict)},
        'ids': ids, 'bf16_logits': bf,
        'other': other,           # round 6: encoder precisions (id agreement, PSNR), the two-rank data-parallel step (gradient / parameter distances, bytes exchanged)
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out['summary']))
    return rc or 0


if __name__ == '__main__':
    sys.exit(main())
