#!/usr/bin/env python3
"""Per-rank compute profile of a destination-range shard on ONE GPU (no exchanges).

Builds rank R of a WORLD-way partition of the benchmark graph exactly as bench.py does, fills the
exchange tables with random rows once (the other ranks' slices would arrive by all-gather), and
times the rank's kernels per class.  What it measures: the compute side of an N-GPU step at its
true shapes (E/N edges against the full source table).  What it cannot: the xGMI exchanges.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


class NoComm:
    native = False
    def all_gather_rows(self, t, w): pass
    def reduce_scatter_rows(self, t, w): pass
    def all_reduce_(self, t): return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--exchange-layer0", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    args = ap.parse_args()
    import torch
    import bench
    pkg = entry.load_package()
    heads, outdims = bench.PRESETS[args.workload]
    n, e, f, c, kind = pkg.synth.SHAPES[args.workload]
    dev = torch.device("cuda", 0)
    dsd = pkg.synth.make_dataset_device(args.workload, dev)        # the device generator: bit-for-bit synth.py, seconds instead of minutes
    row_ptr, col_idx = dsd["row_ptr"], dsd["d_col_idx"].cpu().numpy()
    x_all, lab_all = dsd["d_x"].cpu().numpy(), dsd["d_labels"].cpu().numpy()
    del dsd
    torch.cuda.empty_cache()
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = pkg.GatContext(heads, outdims, f, c, device=0, stream=stream.cuda_stream, collect_timing=True, dtype=args.dtype)
        S = pkg.shard
        plan = S.make_plan(row_ptr, args.world, args.rank)
        rp_l, ci_l = S.local_csr(plan, row_ptr, col_idx)
        lo, hi = plan.row0, plan.row0 + plan.n_rows
        ctx.set_graph(rp_l, ci_l, n_table=plan.n_table, table_row0=plan.table_row0)
        if args.exchange_layer0:
            ctx.set_features(x_all[lo:hi])
        else:
            ctx.set_source_features(plan.table_features(x_all))
        ctx.set_labels(lab_all[lo:hi])
        del x_all
        run = S.ShardedGat(ctx, plan, NoComm(), heads, outdims,
                           alloc=lambda k: torch.randn(k, dtype=torch.float32, device=dev) * 0.1)
        ctx.params_init(42)
        ctx.zero_grad()
        for _ in range(args.warmup):
            run.step()
        ctx.kernel_stats_reset()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        ctx.sync()
        dt = (time.perf_counter() - t0) / args.steps
        stats = ctx.kernel_stats()
    hd = [h * d for h, d in zip(heads, outdims)]
    sb = 2 if args.dtype == "bf16" else 4
    exchanged = range(0 if args.exchange_layer0 else 1, len(heads))
    print(json.dumps({"workload": args.workload, "dtype": args.dtype, "plan": "all layers exchanged" if args.exchange_layer0 else "input replicated, layer 0 exchange-free",
                      "world": args.world, "rank": args.rank, "rows": plan.n_rows, "edges": int(rp_l[-1]),
                      "table_rows": plan.n_table, "ms_per_step_compute_only": dt * 1e3,
                      # bytes every rank RECEIVES per step over xGMI: (world-1)/world of each exchanged table, forward (PL, storage dtype)
                      # and backward (gPL partial sums, fp32), plus the packed-gradient all-reduce (negligible)
                      "wire_MB_per_step_per_rank": sum(plan.n_table * hd[l] * (sb + 4) for l in exchanged) * (args.world - 1) / args.world / 1e6,
                      "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in stats.items() if v[0] > 0}}))
    ctx.close()


if __name__ == "__main__":
    main()
