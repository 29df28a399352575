# Experiment: what would popularity-ordered table rows buy?  The same workloads with the generator's sources
# sorted by popularity (hot rows adjacent: bench.py --sorted-sources) against the benchmark graphs, step time per kernel class and
# the L2 counters of the edge kernels on config 5.    bash tools/sorted_sources_ab.sh <outdir under gpurun_out>
OUT=${1:-sorted}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$OUT
for cfg in "pl10m bf16 5" "products bf16 10" "products f32 10"; do
  set -- $cfg
  for extra in "" "--sorted-sources"; do
    tag=$( [ -z "$extra" ] && echo base || echo sorted )
    python3 bench.py --workload $1 --dtype $2 --steps $3 --warmup 2 --no-cpu-baseline $extra > gpurun_out/$OUT/${1}_${2}_$tag.json 2>/dev/null
    python3 - gpurun_out/$OUT/${1}_${2}_$tag.json "$1 $2 $tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["ms_per_step"],3), d["kernels_ms_per_step"])
PY
  done
done
for extra in "" "--sorted-sources"; do
  tag=$( [ -z "$extra" ] && echo base || echo sorted )
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$OUT/tcc_$tag -- python3 bench.py --workload pl10m --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline $extra > gpurun_out/$OUT/tcc_$tag.json 2> gpurun_out/$OUT/tcc_$tag.err
  echo "== pl10m bf16 $tag"; python3 tools/pmc_summary.py gpurun_out/$OUT/tcc_$tag -kedge_fwd3 -kedge_bwd3
done
