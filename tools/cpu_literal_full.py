#!/usr/bin/env python3
"""ONE full-size step of the literal CPU restatement of the reference (oracle, OpenMP, BASELINE.md §3 flags) on the headline
graph — no sampling, no extrapolation.  Test infrastructure (the oracle is the checker, never the product):
    python tools/cpu_literal_full.py [--workload products] > profiles/r04/cpu_literal_full.json
Prints a heartbeat to stderr every minute (the step is one long native call)."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    import bench
    os.environ["GAT_ORACLE_NATIVE"] = "1"
    pkg = entry.load_package()
    orc = entry.load_oracle()
    heads, outdims = bench.PRESETS[args.workload]
    t0 = time.perf_counter()
    ds = pkg.synth.make_dataset(args.workload, scale=args.scale)
    t_gen = time.perf_counter() - t0
    print(f"graph: {ds['n']} nodes / {ds['e']} edges generated on the host in {t_gen:.1f} s", file=sys.stderr, flush=True)
    cfg = orc.Config(heads, outdims, ds["f"], ds["c"])
    W, a, Wo = orc.xavier_params(cfg, 42)
    stop = threading.Event()

    def beat():
        k = 0
        while not stop.wait(60.0):
            k += 1
            print(f"... literal step running, {k} min", file=sys.stderr, flush=True)
    th = threading.Thread(target=beat, daemon=True)
    th.start()
    t0 = time.perf_counter()
    ref = orc.step(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], ds["x"], W, a, Wo, mt_baseline=True)
    t = time.perf_counter() - t0
    stop.set()
    print(json.dumps({
        "workload": args.workload, "scale": args.scale, "nodes": ds["n"], "edges": ds["e"], "features": ds["f"], "heads": heads, "outdims": outdims,
        "seconds_per_step": t, "edges_per_s": ds["e"] / t, "threads": orc.lib().orc_num_threads(), "cpu_model": bench.cpu_model(),
        "build_flags": orc.build_flags(), "loss_per_node": float(ref.loss_sum_f64) / ds["n"],
        "what": "oracle literal mode (per-edge W.x recomputation, O(sum deg^2) softmax backward, E:279-893), one forward+backward step, "
                "wall time of the native call incl. its result arrays; full graph, no sampling"}), flush=True)


if __name__ == "__main__":
    main()
