#!/bin/bash
# Sweep of the slot-parallel source-major pass (gat_csc.hip "runs") at a shard shape: run length, slots per batch, last-layer
# variant — one shard_profile process per setting (the switches are read once per process).  Output: $OUT/sweep.jsonl
OUT=${OUT:-gpurun_out/r4_sweep}; WORLD=${WORLD:-8}; WL=${WL:-products}; DT=${DT:-f32}
mkdir -p $OUT; : > $OUT/sweep.jsonl
run() {   # label, env...
    local label="$1"; shift
    echo "== $label" >&2
    env "$@" timeout -k 10 150 python tools/shard_profile.py --workload $WL --dtype $DT --world $WORLD --rank 3 --exchange-layer0 2>>$OUT/err.log | sed "s/^{/{\"label\": \"$label\", /" >> $OUT/sweep.jsonl || return 1
}
for s in "$@"; do
    label="$s"; run "$label" $s || exit 1
done
python - <<PY
import json
for l in open("$OUT/sweep.jsonl"):
    d = json.loads(l); k = d["kernels_ms_per_step"]
    print(f'{d["label"]:50s} step {d["ms_per_step_compute_only"]:.3f}  gpl_sum {k.get("gpl_sum", 0):.4f}  edge_bwd {k.get("edge_backward", 0):.3f}')
PY
