#!/bin/bash
# Where the split-K projection's time goes on a small shape (experiment library): kernel time under hipGraph replay with parts of the
# block skipped (GAT_DBG_SPLITK bit mask: 1 A loads, 2 B fill, 4 MFMA loop, 8 slab stores, 16 B fill from constants (no loads), 32 B fill loads only (no cut, no LDS stores)).  tools/splitk_ablate.sh <workload> <outdir>
W=${1:-pubmed}; OUT=${2:-gpurun_out/splitk_ablate}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
export GATV2_LIB=$GRAFT_REPO_ROOT/graph-attention-network-gatv2-_amd/libgatv2_hip_exp.so
for a in ${MASKS:-0 1 2 4 8 3 7 15}; do
  GAT_DBG_SPLITK=$a rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a$a -- python3 bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --graph > /dev/null 2> $OUT/a$a.err
  python3 - $OUT/a$a $a <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'project_splitk_x3' in r['Name'] or 'project_reduce' in r['Name']:
        print('skip mask', sys.argv[2], r['Name'][31:72], round(float(r['AverageNs']) / 1e3, 2), 'us')
PY
done
