#!/usr/bin/env python3
"""Average per-launch counter values per kernel from rocprofv3 --pmc output directories: pmc_table.py <dir> [<dir> ...] [--match substr]"""
import csv, glob, sys, collections, json
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
match = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--match=')), '')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:80]
            if match in k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in acc.items()}
print(json.dumps(out, indent=1))
