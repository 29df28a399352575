#!/usr/bin/env python3
"""Per-kernel time of one step as a function of (nodes, edges) at fixed feature widths: separates the per-row cost
from the per-edge cost of the edge kernels (t ~ a*N + b*E).   python tools/shape_probe.py N E [N E ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry

pkg = entry.load_package()
args = list(map(int, sys.argv[1:]))
for n, e in zip(args[0::2], args[1::2]):
    rp, ci = pkg.synth.powerlaw_graph(n, e)
    x = pkg.synth.features(n, 100); lab = pkg.synth.labels(n, 47)
    with pkg.GatContext([8, 8], [8, 8], 100, 47, collect_timing=True) as ctx:
        ctx.set_graph(rp, ci); ctx.set_features(x); ctx.set_labels(lab); ctx.params_init(42); ctx.zero_grad()
        for _ in range(2):
            ctx.step()
        ctx.kernel_stats_reset()
        for _ in range(5):
            ctx.step()
        st = ctx.kernel_stats()
    print(json.dumps({"n": n, "e": e, "maxdeg": int(np.diff(rp).max()),
                      "ms_per_step": {k: round(v[1] / 5, 3) for k, v in st.items() if v[0]}}), flush=True)
