#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/pmc_traffic.sh -> profiles/<round>/traffic.json (HBM bytes per launch and
kernel class, the figure bench.py attaches as roofline.traffic).

    python tools/make_traffic_json.py gpurun_out/traffic profiles/r01/traffic.json

reads  = 128*TCC_EA0_RDREQ_128B + 64*_64B + 32*_32B   (every read request is 128 B on gfx950: FETCH_SIZE x2,
                                                       MI355X_MICROARCH.md "HBM")
writes = 64*TCC_EA0_WRREQ_64B + 32*(WRREQ - WRREQ_64B)
"""
import collections, csv, glob, json, re, sys

CLASSES = [("edge_last_fused", "edge_last_fused"), ("edge_bwd", "edge_backward"), ("edge_fwd", "edge_forward"), ("gpl_", "gpl_sum"),
           ("gradw_kernel", "grad_w_gemm"), ("gradw_x3_kernel", "grad_w_gemm"), ("EpiProject", "project_gemm"), ("EpiGradX", "grad_x_gemm"),
           ("EpiStore", "grad_x_gemm"), ("head_forward_kernel", "head_forward"), ("head_backward_kernel", "head_backward"),
           ("head_step_kernel", "head_backward")]


def load(d):
    f = (glob.glob(d + "/*/*_counter_collection.csv") or glob.glob(d + "/*_counter_collection.csv")
         or glob.glob(d + "_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    import os
    f.sort(key=os.path.getmtime)                       # several runs in one directory: take the newest
    for r in csv.DictReader(open(f[-1])):
        cls = next((c for pat, c in CLASSES if pat in r["Kernel_Name"]), None)
        if cls is None:
            continue
        agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        if "fix" not in r["Kernel_Name"] and "chunk" not in r["Kernel_Name"]:      # helpers ride with their main kernel
            disp[cls].add(r["Dispatch_Id"])
    return agg, disp


def main(src, dst):
    rd, rdisp = load(src + "/rd")
    wr, wdisp = load(src + "/wr")
    out = {"_comment": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "workload": "products", "kernels": {}}
    for cls in sorted(rd):
        n = max(1, len(rdisp[cls]))
        r = rd[cls]
        reads = 128 * r["TCC_EA0_RDREQ_128B_sum"] + 64 * r["TCC_EA0_RDREQ_64B_sum"] + 32 * r["TCC_EA0_RDREQ_32B_sum"]
        w = wr[cls]
        writes = 64 * w["TCC_EA0_WRREQ_64B_sum"] + 32 * (w["TCC_EA0_WRREQ_sum"] - w["TCC_EA0_WRREQ_64B_sum"])
        nw = max(1, len(wdisp[cls]))
        out["kernels"][cls] = {"read_bytes": reads / n, "write_bytes": writes / nw, "launches_counted": n,
                               "bytes_per_launch": reads / n + writes / nw}
    # the passes run `bench.py --steps 2 --warmup 1`: 3 steps => launches per step, and the step's total HBM traffic
    steps = 3
    tot = 0.0
    for cls, v in out["kernels"].items():
        v["launches_per_step"] = v["launches_counted"] / steps
        tot += v["bytes_per_launch"] * v["launches_per_step"]
    out["bytes_per_step"] = tot
    json.dump(out, open(dst, "w"), indent=1)
    print(f"step total {out['bytes_per_step'] / 1e9:.2f} GB")
    for k, v in out["kernels"].items():
        print(f"{k:16s} {v['bytes_per_launch'] / 1e9:8.3f} GB/launch  (R {v['read_bytes'] / 1e9:.3f} + W {v['write_bytes'] / 1e9:.3f}, n={v['launches_counted']})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
