#!/usr/bin/env python3
"""Where a small-case test process spends its wall time (import, library load, first context, later contexts)."""
import sys, time
t0 = time.time()
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); A = pkg.abi
t1 = time.time(); print("imports + load", round(t1 - t0, 2), flush=True)
from conftest import small_graph
rng = np.random.default_rng(5)
rp, ci = small_graph(rng, 260, 2600, hub=(9, 700), empty=(0, 3))
x = rng.standard_normal((260, 12)).astype(np.float32)
lab = rng.integers(0, 4, 260).astype(np.int32); lab[0] = 3
cfg = orc.Config([8, 8], [8, 8], 12, 4)
W, a, Wo = orc.xavier_params(cfg, 6)
for it in range(4):
    t = time.time()
    ctx = pkg.GatContext([8, 8], [8, 8], 12, 4)
    ta = time.time()
    ctx.set_graph(rp, ci); tb = time.time()
    ctx.set_features(x); ctx.set_labels(lab); tc = time.time()
    for g, arr in enumerate((W, a, Wo)): ctx.params_set(g, arr)
    ctx.zero_grad(); loss, correct = ctx.step(); td = time.time()
    taps = [ctx.tap(A.TAP_PL, l) for l in range(2)]; te = time.time()
    ctx.close(); tf = time.time()
    print(f"ctx {it}: create {ta-t:.2f} set_graph {tb-ta:.2f} features+labels(buffers) {tc-tb:.2f} step {td-tc:.2f} taps {te-td:.2f} close {tf-te:.2f}", flush=True)
