#!/bin/bash
# tools/bench_ab_graph.sh OUTDIR WORKLOAD "ENV1" "ENV2" ... — same-box A/B under hipGraph replay (small, launch-bound shapes)
out=$1; wl=$2; shift 2
mkdir -p "$out"
i=0
for e in "$@"; do
  echo "$i: $e" >> "$out/abg_${wl}_index.txt"
  env $e timeout -k 10 200 python bench.py --workload "$wl" --graph --no-cpu-baseline --steps 300 --warmup 30 > "$out/abg_${wl}_$i.json" 2> "$out/abg_${wl}_$i.err" || { echo "run $i failed"; tail -5 "$out/abg_${wl}_$i.err"; exit 1; }
  python - "$out/abg_${wl}_$i.json" "$e" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2] or "(default)", "->", round(d["ms_per_step"]*1000,1), "us")
PY
  i=$((i+1))
done
