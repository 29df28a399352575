import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
n, e = 2_450_000, 4_000_000
rp, ci = pkg.synth.powerlaw_graph(n, e)
ctx = pkg.GatContext([8, 8], [8, 8], 100, 47, device=0, collect_timing=True)
ctx.set_graph(rp, ci); ctx.set_features(pkg.synth.features(n, 100)); ctx.set_labels(pkg.synth.labels(n, 47))
ctx.params_init(1); ctx.zero_grad()
for _ in range(2): ctx.forward(); ctx.backward()
ctx.kernel_stats_reset()
for _ in range(5): ctx.forward(); ctx.backward()
st = ctx.kernel_stats()
print(os.environ.get("GAT_HEAD_NB"), {k: round(v[1] / 5, 3) for k, v in st.items() if v[0] and k.startswith(("head", "grad", "proj"))}, flush=True)
