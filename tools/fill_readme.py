#!/usr/bin/env python3
"""Regenerate README.md's numbers table (between the numbers:begin / numbers:end markers) from tools/readme_numbers.in and profiles/<tag>_bench_default_line.json (the driver-style default bench line of the final library).
usage: fill_readme.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
d = json.loads(open(os.path.join(ROOT, 'profiles', f'{tag}_bench_default_line.json')).read().strip().splitlines()[-1])
r, sc, t = d['roofline'], d['side_configs'], d['roofline'].get('telemetry') or {}
F = {'HEAD': d['value'], 'FP32': sc[[k for k in sc if k.startswith('fp32_d')][0]]['value'], 'ACH': r['achieved'], 'FRAC': r['frac'], 'FOS': r.get('frac_of_sustained'),
     'PW': t.get('power_w'), 'CLK': t.get('sclk_mhz'), 'PLF': t.get('power_limited_frac'), 'CPU': d.get('cpu_baseline', {}).get('value'),
     'B1': sc['latency_b1_ms']['value'], 'B8': sc['small_batch_b8']['value'], 'B32': sc['small_batch_b32']['value'],
     'D12': sc.get('d12_images_per_s', {}).get('value'), 'D30': sc.get('d30_images_per_s', {}).get('value'),
     'V16': sc['vqvae_roundtrip_b128']['value'], 'VX3': sc['vqvae_roundtrip_b128_bf16x3']['value'], 'V32': sc['vqvae_roundtrip_b128_fp32_encoder']['value'],
     'A16': sc['vqvae_roundtrip_b128'].get('id_agreement_vs_fp32_mode'), 'AX3': sc['vqvae_roundtrip_b128_bf16x3'].get('id_agreement_vs_fp32_mode'),
     'A32': sc['vqvae_roundtrip_b128_fp32_encoder'].get('id_agreement_vs_fp32_mode'),
     'TR': sc[[k for k in sc if k.startswith('train_d') and not k.endswith('labels')][0]]['value'], 'TRX': sc[[k for k in sc if k.endswith('bf16x3_labels')][0]]['value']}
p = os.path.join(ROOT, 'README.md')
s = open(p).read()
tab = open(os.path.join(ROOT, 'tools', 'readme_numbers.in')).read()
for k, v in F.items():
    tab = tab.replace(f'@{k}@', str(v))
left = re.findall(r'@[A-Z0-9]+@', tab)
b, e = s.index('<!-- numbers:begin'), s.index('<!-- numbers:end -->')
s = s[:s.index('\n', b) + 1] + tab + s[e:]
open(p, 'w').write(s)
print('filled', len(F), 'fields; unfilled:', left)
