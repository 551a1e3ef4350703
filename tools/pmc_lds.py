#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass: conflict cycles / LDS-array cycles."""
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:90]][r['Counter_Name']] += float(r['Counter_Value'])
out = {}
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0.0)):
    act, conf = d.get('SQ_LDS_IDX_ACTIVE', 0.0), d.get('SQ_LDS_BANK_CONFLICT', 0.0)
    if act > 0:
        out[k] = {'lds_active_cycles': act, 'bank_conflict_cycles': conf, 'conflict_share': round(conf / act, 4)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
