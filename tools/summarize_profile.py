#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into one small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    lines.append(f'# {f}')
    lines += [ln.rstrip() for ln in open(f)][:12]
for name in ('fetch', 'write'):
    for f in glob.glob(os.path.join(out, name, '**', '*counter_collection.csv'), recursive=True):
        agg = defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            key = (row['Kernel_Name'].split('(')[0][:60], row['Counter_Name'])
            agg[key][0] += 1
            agg[key][1] += float(row['Counter_Value'])
        lines.append(f'# {f}: per-dispatch mean of each counter (units as reported by rocprofv3)')
        for (k, c), (n, s) in sorted(agg.items()):
            lines.append(f'{k:60s} {c:12s} dispatches={n:4d} mean={s / n:.6g}')
for f in glob.glob(os.path.join(out, 'bench_*.log')):
    for ln in open(f):
        if ln.startswith('{"metric"'):
            lines.append(f'# {os.path.basename(f)}: {ln.strip()}')
text = '\n'.join(lines)
open(os.path.join(out, 'summary.txt'), 'w').write(text + '\n')
print(text)
