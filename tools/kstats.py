#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [steps]  (steps: divide calls to get launches per step)"""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    name = re.sub(r'\(anonymous namespace\)::|void |gat::', '', r['Name'])
    name = re.sub(r'\(.*', '', name)[:70]
    per = f"{int(r['Calls'])/steps:5.1f}/step" if steps else f"{r['Calls']:>7s}"
    print(f"{name:70s} {per} {float(r['AverageNs'])/1e3:9.2f} us {100*float(r['TotalDurationNs'])/tot:5.1f} %")
