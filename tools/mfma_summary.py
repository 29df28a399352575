#!/usr/bin/env python3
"""Matrix-pipe / VALU / wait shares per kernel from one rocprofv3 --kernel-trace --pmc pass (tools/collect_profiles_r3.sh, part
"mfma"):  python tools/mfma_summary.py <dir> [name filter ...]

Per dispatch: duration from the kernel trace; cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs, MI355X_MICROARCH.md);
  mfma  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)           (cycles: 32 per v_mfma_f32_32x32x16_bf16)
  valu  = 4 x SQ_ACTIVE_INST_VALU / (1024 x cycles)                   (quad-cycles -> cycles)
  wait  = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                (share of wave time parked on s_waitcnt / barriers)
Dispatches of one kernel are grouped by grid size (the layers of the model differ in shape)."""
import collections, csv, glob, re, sys
d = sys.argv[1]; filt = sys.argv[2:]
cc = glob.glob(d + '/*/*_counter_collection.csv')[0]
kt = glob.glob(d + '/*/*_kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
per = collections.defaultdict(dict); meta = {}
for r in csv.DictReader(open(cc)):
    per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    meta[r['Dispatch_Id']] = (re.sub(r'\(anonymous namespace\)::|void |gat::', '', r['Kernel_Name']).split('(')[0], r.get('Grid_Size', '?'))
groups = collections.defaultdict(list)
for did, c in per.items():
    groups[meta[did]].append((dur.get(did, 0.0), c))
print(f"{'kernel':74s} {'grid':>9s} {'n':>3s} {'us':>8s} {'mfma':>6s} {'valu':>6s} {'wait':>6s}")
for (name, grid), items in sorted(groups.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    if filt and not any(f in name for f in filt):
        continue
    n = len(items)
    us = sum(x[0] for x in items) / n
    g = lambda k: sum(x[1].get(k, 0.0) for x in items) / n
    cyc = g('GRBM_GUI_ACTIVE') / 8.0
    if cyc <= 0 or us < 3.0:
        continue
    wc = g('SQ_WAVE_CYCLES')
    print(f"{name[:74]:74s} {grid:>9s} {n:3d} {us:8.1f} {g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):6.1%} {4 * g('SQ_ACTIVE_INST_VALU') / (1024 * cyc):6.1%} "
          f"{(g('SQ_WAIT_ANY') / wc if wc else 0):6.1%}")
