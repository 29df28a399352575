#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, re, sys
def load(d):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::|void |gat::', '', r['Kernel_Name']).split('(')[0][:48]
        agg[name][r['Counter_Name']] += float(r['Counter_Value']); disp[name].add(r['Dispatch_Id'])
    return agg, disp
filt = [a[2:] for a in sys.argv[1:] if a.startswith('-k')]
for d in [a for a in sys.argv[1:] if not a.startswith('-k')]:
    agg, disp = load(d)
    print("==", d)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
        if filt and not any(f in k for f in filt): continue
        n = len(disp[k])
        print(" ", k, 'n=%d' % n)
        for c, val in sorted(v.items()): print("      %-36s per-dispatch %.4g" % (c, val / n))
