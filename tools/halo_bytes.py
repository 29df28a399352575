#!/usr/bin/env python3
"""Bytes on the wire per rank and step, full exchange vs halo (GAT_COMM_HALO), for the benchmark graphs.

Computed from the generator's graph (device generator, bit-for-bit synth.py), no exchange is run: for every rank of a P-way
destination-range partition, the rows of every OTHER rank's slice that its edges reference.  Full exchange: each rank receives
(P-1) x max_rows rows per exchanged table; halo: the referenced rows.  Tables per step: forward PL (storage dtype) + backward
gPL (fp32) for every exchanged layer.  python tools/halo_bytes.py [--workload products|pl10m] > profiles/r04/halo_bytes_<w>.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--worlds", default="2,4,8")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    pkg = entry.load_package()
    heads, outdims = bench.PRESETS[args.workload]
    dtype = args.dtype or ("bf16" if args.workload == "pl10m" else "f32")
    sb = 2 if dtype == "bf16" else 4
    hd = [h * d for h, d in zip(heads, outdims)]
    dev = torch.device("cuda", 0)
    dsd = pkg.synth.make_dataset_device(args.workload, dev)
    row_ptr = np.asarray(dsd["row_ptr"])
    col = dsd["d_col_idx"]                                    # device int32 [E], global source ids
    n = len(row_ptr) - 1
    out = {"workload": args.workload, "dtype": dtype, "nodes": n, "edges": int(row_ptr[-1]), "note": "computed from the graph; no exchange was run (unmeasured on hardware)", "worlds": []}
    S = pkg.shard
    for P in [int(w) for w in args.worlds.split(",")]:
        plans = [S.make_plan(row_ptr, P, r) for r in range(P)]
        bounds = plans[0].bounds
        max_rows = plans[0].n_table // P
        owner_lo = torch.tensor(bounds[:-1], device=dev, dtype=torch.int64)
        per_rank = []
        for r in range(P):
            e0, e1 = int(row_ptr[bounds[r]]), int(row_ptr[bounds[r + 1]])
            u = torch.unique(col[e0:e1].to(torch.int64))
            own = torch.bucketize(u, owner_lo, right=True) - 1          # owner rank of every referenced node
            remote = int((own != r).sum())
            per_rank.append({"rank": r, "edges": e1 - e0, "referenced_remote_rows": remote, "full_remote_rows": (P - 1) * max_rows,
                             "fraction": remote / ((P - 1) * max_rows)})
        frac = sum(x["referenced_remote_rows"] for x in per_rank) / sum(x["full_remote_rows"] for x in per_rank)
        for plan_name, layers in (("input replicated, layer 0 exchange-free", range(1, len(hd))), ("all layers exchanged", range(len(hd)))):
            row_bytes = sum(hd[l] * (sb + 4) for l in layers)       # forward PL row + backward gPL row, per exchanged layer
            full = max(x["full_remote_rows"] for x in per_rank) * row_bytes
            halo = max(x["referenced_remote_rows"] for x in per_rank) * row_bytes
            out["worlds"].append({"world": P, "plan": plan_name, "max_rows": max_rows, "referenced_fraction": frac,
                                  "wire_MB_per_rank_full": full / 1e6, "wire_MB_per_rank_halo_max": halo / 1e6,
                                  "halo_default": "on" if frac < 0.5 else "off (GAT_COMM_HALO=2 rule: fraction >= 0.5)"})
        out["worlds"][-1]["per_rank"] = per_rank
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
