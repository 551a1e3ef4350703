#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean FETCH_SIZE / WRITE_SIZE per launch of the GEMM kernel family.
Units/corrections as MI355X_MICROARCH.md section HBM prescribes: the counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import csv, glob, json, sys, collections
out = {}
for tag in ('fetch', 'write'):
    f = glob.glob(f'gpurun_out/pmc_{tag}/**/*counter_collection.csv', recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'cvar_gemm_kernel' in r['Kernel_Name'] or 'conv3x3_halo' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        out[k] = dict(mean=sum(v) / len(v), launches=len(v))
fetch = out.get('FETCH_SIZE', {}).get('mean', 0.0) * 1024 * 2
write = out.get('WRITE_SIZE', {}).get('mean', 0.0) * 1024
label = sys.argv[1] if len(sys.argv) > 1 else 'one d24 generation'
res = dict(kernel=f'cvar_gemm_kernel + conv3x3_halo_bf16_kernel (all launches of {label})', fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
           bytes_per_launch=fetch + write, launches=out.get('FETCH_SIZE', {}).get('launches'), note='FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)',
           collected=f'separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/final_measure.sh) of {label}; mean over the GEMM-family launches')
try:      # stamp: the library the counters were collected on (digest of the kernel sources; .git does not travel to the GPU box)
    import os
    res['lib_digest'] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'controlvar_amd', 'csrc', 'build', 'digest.txt')).read().strip()[:16]
except OSError:
    res['lib_digest'] = None
print(json.dumps(res))
json.dump(res, open('gpurun_out/gemm_hbm_traffic.json', 'w'))
