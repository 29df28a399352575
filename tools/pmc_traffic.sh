# HBM-side traffic of every kernel of one bench step: exact request-size counters of the L2's
# memory-side (EA) interface, in separate passes (MI355X_MICROARCH.md §HBM; TCC has 4 slots).
#   bash tools/pmc_traffic.sh [outdir under gpurun_out, default traffic] [extra bench.py args...]
OUT=${1:-traffic}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$OUT/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/$OUT/$name.json 2> gpurun_out/$OUT/$name.err; }
EXTRA="$*"
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
ls gpurun_out/$OUT
