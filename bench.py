#!/usr/bin/env python3
"""bench.py — edges/sec of one GATv2 forward+backward step (hot path of GATv2_edge_based.cu) on
N MI355X, on the OGBN-Products-shape synthetic power-law graph BASELINE.json quotes the metric on.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward over all layers + output head + loss (scalars read back) + backward over all
layers; optimizer / zero-grad excluded.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel class, timed live with
HIP events on the context's stream inside the timed steps; `cpu_baseline` is the literal CPU
restatement of the reference's algorithm (oracle/) on a bounded sample of the same graph law.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

PRESETS = {   # workload -> (heads, outdims); BASELINE.json gives layer/head counts, SURVEY §8 fixes outdims
    "products": ([8, 8], [8, 8]),
    "pubmed": ([8, 8], [8, 8]),
    "cora": ([8, 8], [8, 8]),
    "arxiv": ([8, 8, 8], [8, 8, 8]),
    "pl10m": ([4, 4], [8, 8]),          # BASELINE config 5 (10 M nodes / 250 M edges, 4 heads; run with --dtype bf16)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="products", choices=sorted(PRESETS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink nodes and edges (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange-layer0", action="store_true",
                    help="N>1: do not replicate the input features; all-gather / reduce-scatter layer 0 too (A/B)")
    ap.add_argument("--cpu-sample-scale", type=float, default=0.0, help="0 = auto (~15 s of CPU work)")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: replay the step as one hipGraph launch (launch-bound small graphs: cora / pubmed / "
                         "arxiv); per-kernel event timing, hence the roofline object, is not available in this mode")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the gathered / exchanged PL table and the message rows (BASELINE config 5 "
                         "is bf16); arithmetic is fp32 either way.  The headline metric is f32.")
    ap.add_argument("--comm", choices=["native", "torch"], default="native",
                    help="N>1 exchanges: 'native' = RCCL called by the library on the context's stream (in-place "
                         "all-gather / reduce-scatter, one all-reduce); 'torch' = the same plan driven from "
                         "shard.ShardedGat over torch.distributed")
    ap.add_argument("--comm-chunks", type=int, default=1,
                    help="N>1, --comm native: chunk-pipelined forward exchange (gat_comm_option GAT_COMM_PIPELINE); results unchanged.  "
                         "UNVERIFIED over RCCL beyond world 1 (no multi-GPU box in the build pool): default off")
    ap.add_argument("--gpl-bf16", action="store_true",
                    help="N>1, --comm native: remote gPL partial sums travel as bf16 (GAT_COMM_GPL_BF16; 1e-2 parity mode, NOT the headline).  "
                         "UNVERIFIED over RCCL beyond world 1 (only its host-transport twin is tested at 3 ranks): default off")
    ap.add_argument("--halo", type=int, default=0, choices=[0, 1, 2],
                    help="N>1, --comm native: table exchanges move only the rows the receiving shard's edges reference (gat_comm_option "
                         "GAT_COMM_HALO; 2 = only if fewer than half of the rows would travel).  Same results as the full exchange.  "
                         "UNVERIFIED over RCCL beyond world 1: default off")
    ap.add_argument("--beta", type=float, default=0.75, help="power-law exponent of the degree law (debug)")
    ap.add_argument("--sorted-sources", action="store_true",
                    help="experiment, NOT the benchmark graph: source popularity decreasing with the node id (hot table rows adjacent) — "
                         "what a popularity ordering of the table rows inside the library would buy (DESIGN 4, config 5)")
    return ap.parse_args()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pkg, workload, heads, outdims, sample_scale):
    """Literal reference algorithm (per-edge W·x recomputation, O(deg^2) softmax backward) on the
    host cores, on a scaled-down graph of the same law.  Reported, never the target."""
    os.environ["GAT_ORACLE_NATIVE"] = "1"      # BASELINE.md §3 flags, compiled on this machine (oracle/Makefile)
    orc = entry.load_oracle()
    n_full, e_full, f, c, kind = pkg.synth.SHAPES[workload]

    def run(scale):
        ds = pkg.synth.make_dataset(workload, scale=scale)
        cfg = orc.Config(heads, outdims, ds["f"], ds["c"])
        W, a, Wo = orc.xavier_params(cfg, 42)
        t0 = time.perf_counter()
        orc.step(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], ds["x"], W, a, Wo, mt_baseline=True)
        return ds, time.perf_counter() - t0

    if sample_scale <= 0:
        probe_scale = min(1.0, max(2000.0 / e_full, 1e-4))
        ds, t = run(probe_scale)                       # calibrate on ~2k edges
        per_edge = max(t / ds["e"], 1e-9)
        sample_scale = min(1.0, 30.0 / (per_edge * e_full))    # the 2k-edge probe over-estimates per-edge cost (threads idle)
    ds, t = run(sample_scale)
    if t < 8.0 and sample_scale < 1.0:       # aim at 10-30 s of CPU work (contract): rescale once from the measured rate
        sample_scale = min(1.0, sample_scale * 14.0 / max(t, 1e-3))
        ds, t = run(sample_scale)
    return {
        "value": ds["e"] / t, "unit": "edges/s", "cores": orc.lib().orc_num_threads(), "kind": "port",
        "cpu_model": cpu_model(), "build_flags": orc.build_flags(),
        "host_cpus": f"{os.cpu_count()} logical CPUs visible; OpenMP team = the CPUs this job may use (affinity capped by the cgroup quota): {orc.effective_cpus()}",
        "sample": f"{workload}-law graph scaled to {ds['n']} nodes / {ds['e']} edges / {ds['f']} feat, "
                  f"1 step fwd+bwd in {t:.2f} s (oracle literal mode, OpenMP); the literal algorithm's softmax backward is "
                  f"O(sum deg^2), which grows faster than E: the full graph would run at FEWER edges/s than this sample",
    }


def cpu_baseline_cora_shape(pkg):
    """the Cora-shape config (BASELINE.json configs[0], "CPU reference path") at FULL size — no sampling, no
    extrapolation: literal reference algorithm, OpenMP, one fwd+bwd step (best of 3)."""
    os.environ["GAT_ORACLE_NATIVE"] = "1"
    orc = entry.load_oracle()
    heads, outdims = PRESETS["cora"]
    ds = pkg.synth.make_dataset("cora")
    cfg = orc.Config(heads, outdims, ds["f"], ds["c"])
    W, a, Wo = orc.xavier_params(cfg, 42)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        orc.step(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], ds["x"], W, a, Wo, mt_baseline=True)
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return {"value": ds["e"] / best, "unit": "edges/s", "cores": orc.lib().orc_num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"cora-shape at full size ({ds['n']} nodes / {ds['e']} edges / {ds['f']} feat), 1 step fwd+bwd in {best * 1e3:.1f} ms "
                      f"(oracle literal mode, OpenMP, best of 3)"}


def cpu_baseline_restructured(pkg, workload, heads, outdims):
    """Second CPU line: the RESTRUCTURED algorithm — the one the HIP kernels implement — on the
    host cores (oracle/gatv2_oracle.cpp::orc_step_restructured, OpenMP), on a same-law sample sized for
    about 10 s.  A fairer comparison than the literal reference algorithm; still only a reported baseline."""
    orc = entry.load_oracle()
    n_full, e_full, f, c, kind = pkg.synth.SHAPES[workload]

    def run(scale):
        ds = pkg.synth.make_dataset(workload, scale=scale)
        cfg = orc.Config(heads, outdims, ds["f"], ds["c"])
        W, a, Wo = orc.xavier_params(cfg, 42)
        t0 = time.perf_counter()
        orc.step_restructured(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], ds["x"], W, a, Wo)
        return ds, time.perf_counter() - t0

    ds, t = run(min(1.0, max(200000.0 / e_full, 1e-3)))          # calibrate on ~200k edges
    scale = min(1.0, 0.5, 25.0 * ds["e"] / (max(t, 1e-6) * e_full))
    if scale * e_full > 2 * ds["e"]:
        ds, t = run(scale)
    return {
        "value": ds["e"] / t, "unit": "edges/s", "cores": orc.lib().orc_num_threads(), "kind": "port",
        "cpu_model": cpu_model(), "build_flags": orc.build_flags(),
        "sample": f"{workload}-law graph scaled to {ds['n']} nodes / {ds['e']} edges / {ds['f']} feat, 1 step fwd+bwd in "
                  f"{t:.2f} s (restructured algorithm = the HIP path's, OpenMP; not the reference's kernels)",
    }


def measured_traffic(args, world, kernel):
    """HBM bytes per launch of `kernel` (or, kernel=None, per whole step) from the committed rocprofv3 PMC passes
    (profiles/*/traffic.json, collected with tools/pmc_traffic.sh on this workload at N=1; the newest round wins):
    PMC counters cannot be read from inside the benchmark process, so the figure is attached only when workload,
    dtype and scale match, else null."""
    return measured_traffic_src(args, world, kernel)[0]


def measured_traffic_src(args, world, kernel):
    """(bytes, "profiles/rNN/traffic.json (rocprofv3 PMC, builder-run)") — the figure and the committed file it was read from: it
    is NOT measured in this run (a reader of the JSON line must be able to tell: VERDICT r3)."""
    if world != 1 or args.scale != 1.0:
        return None, None
    best, src = None, None
    for path in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "*", "traffic*.json"))):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("workload") != args.workload or t.get("dtype", "f32") != args.dtype:
            continue
        if kernel is None:
            if t.get("bytes_per_step") is not None:
                best, src = t["bytes_per_step"], path
        elif kernel in t.get("kernels", {}):
            best, src = t["kernels"][kernel]["bytes_per_launch"], path
    if src is not None:
        src = os.path.relpath(src, ROOT) + " (rocprofv3 PMC passes run by the builder with tools/pmc_traffic.sh; a committed constant, not measured in this run)"
    return best, src


def main():
    args = parse()
    # native libraries (RCCL's version banner) write to fd 1: keep stdout for the one JSON line
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    pkg = entry.load_package()
    A = pkg.abi
    heads, outdims = PRESETS[args.workload]
    n, e, f, c, kind = pkg.synth.SHAPES[args.workload]
    if args.scale != 1.0:
        n, e = max(16, int(n * args.scale)), max(16, int(e * args.scale))

    if os.environ.get("GAT_BENCH_SINGLE_DEVICE") == "1":      # rehearsal: all ranks share GPU 0 (use --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, **kw)

    # ---- synthetic inputs (same on every rank; each keeps its destination range) ----
    # generated ON the device (csrc/gat_synth.hip; bit-for-bit the host generator's arrays): the host only builds the
    # N-sized tables.  N > 1: every rank draws the whole graph on its own GPU and keeps its destination range.
    t_gen = time.perf_counter()
    dsd = pkg.synth.make_dataset_device(args.workload, dev, scale=args.scale, beta=args.beta, sorted_sources=args.sorted_sources)
    row_ptr = dsd["row_ptr"]
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        use_graph = args.graph and world == 1 and os.environ.get("GAT_FORCE_SHARDED") != "1"
        ctx = pkg.GatContext(heads, outdims, f, c, device=local_rank, stream=stream.cuda_stream,
                             collect_timing=not use_graph, dtype=args.dtype)
        force_sharded = os.environ.get("GAT_FORCE_SHARDED") == "1"     # rehearse the N>1 code path on one GPU
        if world == 1 and force_sharded:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group(backend=args.backend, rank=0, world_size=1)
        t_up = 0.0
        if world == 1 and not force_sharded:
            t_up = time.perf_counter()
            d_rp = torch.from_numpy(row_ptr).to(dev)
            ctx.set_graph_device(d_rp.data_ptr(), dsd["d_col_idx"].data_ptr(), n, e)      # + work list + source-major index
            ctx.set_features_device(dsd["d_x"].data_ptr(), n, f)
            ctx.set_labels_device(dsd["d_labels"].data_ptr(), n)
            ctx.sync()
            t_up = time.perf_counter() - t_up
            del d_rp
            runner = None
        else:
            col_idx = dsd["d_col_idx"].cpu().numpy()
            S = pkg.shard
            plan = S.make_plan(row_ptr, world, rank)
            rp_l, ci_l = S.local_csr(plan, row_ptr, col_idx)
            lo, hi = plan.row0, plan.row0 + plan.n_rows
            ctx.set_graph(rp_l, ci_l, n_table=plan.n_table, table_row0=plan.table_row0)
            if args.exchange_layer0:     # A/B: exchange layer 0 like the hidden layers
                ctx.set_features_device(dsd["d_x"][lo:hi].contiguous().data_ptr(), hi - lo, f)
            else:                        # static input replicated on every rank: layer 0 runs without exchanges
                ctx.set_source_features(plan.table_features(dsd["d_x"].cpu().numpy()))
            ctx.set_labels_device(dsd["d_labels"][lo:hi].contiguous().data_ptr(), hi - lo)
            del col_idx
            comm_kind = args.comm if args.backend == "nccl" else "torch"
            if comm_kind == "native":
                # the library's own RCCL communicator; torch.distributed only ships the 128-byte id
                idt = torch.zeros(A.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(pkg.GatContext.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                ctx.comm_init_rccl(world, rank, bytes(idt.cpu().tolist()))
                if args.comm_chunks > 1:
                    ctx.comm_option(A.COMM_PIPELINE, args.comm_chunks)
                if args.gpl_bf16:
                    ctx.comm_option(A.COMM_GPL_BF16, 1)
                if args.halo:
                    ctx.comm_option(A.COMM_HALO, args.halo)
                runner = ctx
            else:
                tcomm = S.TorchComm()
                if args.halo:                       # torch.distributed twin of GAT_COMM_HALO (value 2: by the referenced fraction)
                    frac = tcomm.halo_setup(plan, ci_l)
                runner = S.ShardedGat(ctx, plan, tcomm, heads, outdims,
                                      alloc=lambda k: torch.empty(k, dtype=torch.float32, device=dev),
                                      halo=bool(args.halo == 1 or (args.halo == 2 and frac < 0.5)))
        del row_ptr, dsd
        torch.cuda.empty_cache()
        ctx.params_init(42)
        ctx.zero_grad()
        t_gen = time.perf_counter() - t_gen

        if use_graph:
            ctx.step_graph(True)

        def step():
            if use_graph:
                out = ctx.step()
            elif runner is None:
                out = ctx.step()             # forward + head + loss + backward, loss / #correct read back once at the end
            else:
                out = runner.step()
            return out

        for _ in range(args.warmup):
            step()
        ctx.zero_grad()
        ctx.kernel_stats_reset()

        def fence():
            ctx.sync()
            torch.cuda.synchronize(dev)
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize(dev)

        # per-step HIP events on the context's stream (median of the timed steps); recorded without a
        # host sync, read after the closing fence.  `value` stays the wall-clock total of the bracketed region.
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record(stream)
            loss, correct = step()
            ev[i][1].record(stream)
        fence()
        dt = time.perf_counter() - t0
        step_ms = sorted(a.elapsed_time(b) for a, b in ev)
        median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
        stats = ctx.kernel_stats()
        bytes_step, bytes_k = ctx.algorithmic_bytes()

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        bt = torch.tensor([bytes_step], dtype=torch.float64, device=dev)
        dist.all_reduce(bt)
        bytes_step_all = float(bt.item())
    else:
        bytes_step_all = bytes_step

    if rank == 0:
        ms = dt / args.steps * 1e3
        step_traffic = measured_traffic(args, world, None)
        req_bytes_step = A.request_bytes_shape(heads, outdims, f, c, n, e, dtype=args.dtype)[0] if world == 1 and args.scale == 1.0 else None
        timed = [k for k in stats if stats[k][0] > 0 and k not in ("misc", "exchange")]
        if timed:
            dom = max(timed, key=lambda k: stats[k][1])
            launches, tot_ms = stats[dom]
            per_launch_bytes = bytes_k[dom] * args.steps / launches
            avg_ms = tot_ms / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        # SURVEY 8d's byte count prices an attn_coeff write (forward) and read (backward) of E*H elements per layer that the
        # training path never performs (alpha is recomputed from the per-row softmax statistics): the same quotients on the
        # bytes this path's algorithm really needs are reported beside the contract's `frac` — never instead of it
        b_store = 4 if args.dtype == "f32" else 2
        alpha_dir = float(sum(e * h * b_store for h in heads)) if runner is None else 0.0
        tp_bytes_step = bytes_step_all - 2.0 * alpha_dir if runner is None else None
        pmc_note = ("traffic = L2-to-fabric request bytes (TCC_EA_RDREQ / WRREQ, rocprofv3 PMC): Infinity-Cache hits are counted in it "
                    "(MI355X_MICROARCH.md, HBM), so it is an upper bound of DRAM bytes, and `frac` / `achieved` are fractions of SURVEY 8d's "
                    "byte model (which assumes every gathered row misses), not a measured DRAM utilisation")
        line = {
            "metric": "edges/sec (fwd+bwd, 2-layer 8-head GATv2)" if len(heads) == 2 else
                      f"edges/sec (fwd+bwd, {len(heads)}-layer 8-head GATv2)",
            "value": e / (dt / args.steps), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "ms_per_step_median_hipevents": median_ms,
            "ms_per_step_min_max_hipevents": [step_ms[0], step_ms[-1]], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32" if args.dtype == "f32" else "f32 arithmetic, bf16 PL/message storage",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-shape synthetic power-law graph, {n} nodes / {e} edges / {f} feat / "
                            f"{c} classes, {len(heads)}-layer GATv2 heads {heads} outdims {outdims}, "
                            + ("fp32" if args.dtype == "f32" else "fp32 arithmetic / bf16 PL+message storage"),
                "launch": "hipGraph replay" if use_graph else "eager",
                "parallelism": (f"dst-range x{world}, " + ("all layers exchanged" if args.exchange_layer0 else
                                "input features replicated (layer 0 exchange-free)") + f", exchanges: {comm_kind}"
                                + (f", forward exchange pipelined in {args.comm_chunks} chunks" if args.comm_chunks > 1 else "")
                                + (", remote gPL partials as bf16" if args.gpl_bf16 else "")
                                + (f", halo exchange (active={ctx.comm_halo_info()[0]}, referenced fraction {ctx.comm_halo_info()[3]:.3f})" if args.halo and comm_kind == "native" else ""))
                               if runner is not None else "single GPU",
                "loss_per_node": loss / n, "setup_s": round(t_gen, 1), "index_s": round(t_up, 2),
                "generator": "device (csrc/gat_synth.hip), bit-for-bit synth.py" + (" — EXPERIMENT: sources sorted by popularity" if args.sorted_sources else ""),
            },
            "step_roofline": {"algorithmic_GB_per_step": bytes_step_all / 1e9,
                              "achieved_GBps": bytes_step_all / (dt / args.steps) / 1e9 / world,
                              "frac_of_8TBps_per_gpu": bytes_step_all / (dt / args.steps) / 1e9 / world / HBM_PEAK_GBS,
                              # HBM bytes the step really moves (sum of the PMC passes over all its kernels), and its
                              # ratio to the algorithmic bytes: > 1 = traffic the algorithm does not need
                              "traffic": step_traffic, "traffic_source": measured_traffic_src(args, world, None)[1],
                              "traffic_over_algorithmic": (step_traffic / bytes_step_all) if step_traffic else None,
                              # the same model with every per-edge gathered / scattered row rounded up to whole 128-byte fabric requests
                              # (gat_request_bytes_shape): what the memory system can serve — differs from the figure above for bf16 rows at
                              # H*D < 64 (config 5: 64-byte rows cost a line each); single GPU only
                              "request_granular_GB_per_step": (req_bytes_step / 1e9) if req_bytes_step else None,
                              "frac_request_granular": (req_bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS) if req_bytes_step else None,
                              "training_path_GB_per_step": (tp_bytes_step / 1e9) if tp_bytes_step else None,
                              "frac_training_path": (tp_bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS) if tp_bytes_step else None,
                              "note": pmc_note},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, world, dom),
                         "traffic_source": measured_traffic_src(args, world, dom)[1],
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": per_launch_bytes,
                         # the kernel's REAL HBM rate (PMC bytes / measured time): what the memory system delivers to it,
                         # whatever share of those bytes the algorithm needs
                         "achieved_traffic_GBps": (measured_traffic(args, world, dom) / (avg_ms * 1e-3) / 1e9)
                                                  if measured_traffic(args, world, dom) else None,
                         # the same kernel on the bytes the training path needs (no attn_coeff read / write: see above)
                         "frac_training_path": ((per_launch_bytes - alpha_dir * args.steps / launches) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                               if (runner is None and dom in ("edge_forward", "edge_backward")) else None,
                         "note": pmc_note} if timed else None,
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in stats.items() if v[0] > 0},
        }
        if not args.no_cpu_baseline and world == 1:      # CPU lines: rank 0 at N = 1 only (the other ranks would wait in the barrier)
            line["cpu_baseline"] = cpu_baseline(pkg, args.workload, heads, outdims, args.cpu_sample_scale)
            line["cpu_baseline_restructured"] = cpu_baseline_restructured(pkg, args.workload, heads, outdims)
            line["cpu_baseline_cora_shape"] = cpu_baseline_cora_shape(pkg)          # "Cora-shape always"
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
