"""Experiment: per-class kernel time of one step at fixed E = 61.9 M edges while the node count (hence the\nsize of the gathered PL table) varies — does a table that fits the 256 MB Infinity Cache gather faster?"""
import sys, os, json, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as entry
pkg = entry.load_package()
E = 61_900_000
for n in (200_000, 400_000, 800_000, 1_600_000, 2_450_000):
    rp, ci = pkg.synth.powerlaw_graph(n, E)
    ctx = pkg.GatContext([8, 8], [8, 8], 100, 47, device=0, collect_timing=True)
    ctx.set_graph(rp, ci); ctx.set_features(pkg.synth.features(n, 100)); ctx.set_labels(pkg.synth.labels(n, 47))
    ctx.params_init(1); ctx.zero_grad()
    for _ in range(2): ctx.forward(); ctx.backward()
    ctx.kernel_stats_reset()
    for _ in range(5): ctx.forward(); ctx.backward()
    st = ctx.kernel_stats()
    print(json.dumps({"n": n, "table_MB": n * 256 / 1e6, **{k: round(v[1] / 5, 3) for k, v in st.items() if v[0]}}), flush=True)
    ctx.close()
