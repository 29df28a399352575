#!/bin/bash
# tools/bench_ab.sh OUTDIR WORKLOAD DTYPE "ENV1" "ENV2" ... — same-box A/B of bench.py under different env settings
# (each setting once, in order; one JSON line per run in OUTDIR/ab_<i>.json, the settings in OUTDIR/ab_index.txt)
out=$1; wl=$2; dt=$3; shift 3
mkdir -p "$out"
i=0
root=$(cd "$(dirname "$0")/.." && pwd)
for e in "$@"; do
  echo "$i: $e" >> "$out/ab_index.txt"
  # GAT_DBG (timing-only / wrong-result experiments) exists only in the experiment library
  case "$e" in *GAT_DBG*) e="$e GATV2_LIB=$root/graph-attention-network-gatv2-_amd/libgatv2_hip_exp.so";; esac
  env $e timeout -k 10 300 python bench.py --workload "$wl" --dtype "$dt" --no-cpu-baseline --steps ${STEPS:-10} --warmup 3 > "$out/ab_$i.json" 2> "$out/ab_$i.err" || { echo "run $i failed"; tail -5 "$out/ab_$i.err"; exit 1; }
  python - "$out/ab_$i.json" "$e" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2] or "(default)", "->", round(d["ms_per_step"],3), "ms", d.get("kernels_ms_per_step"))
PY
  i=$((i+1))
done
