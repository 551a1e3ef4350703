#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (counter_collection CSV) per kernel family:
MFMA utilisation = MFMA-busy cycles / (active cycles per XCD x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv, glob, json, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_mfma'
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(src + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        fam = ('cvar_gemm_kernel 256x256' if 'cvar_gemm_kernel' in name and ', 256, 256,' in name else
               'cvar_gemm_kernel conv 256x160' if 'cvar_gemm_kernel' in name and ', 256, 160,' in name else
               'cvar_gemm_kernel other' if 'cvar_gemm_kernel' in name else
               'conv3x3_halo_bf16_kernel' if 'conv3x3_halo' in name else
               'attn_mfma_bf16_kernel' if 'attn_mfma_bf16' in name else 'other')
        acc[fam][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            calls[fam] += 1
out = {}
for fam, d in acc.items():
    gui = d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    out[fam] = {'launches': calls[fam], 'active_cycles_per_xcd': gui, 'mfma_busy_cycles_all_simds': busy,
                'mfma_utilisation': round(busy / (gui * 1024.0), 4) if gui else None}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/pmc_mfma_util.json', 'w'), indent=1)
