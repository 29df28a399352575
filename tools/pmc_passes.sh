# SQ / TLB / L2 counter passes of one bench step (separate rocprofv3 --pmc runs).
#   bash tools/pmc_passes.sh [outdir under gpurun_out, default pmc2] [passes, default "sq tlb tcc lat"]
#   PMC_PROG="tools/shard_profile.py --world 8 --rank 3 --exchange-layer0" profiles another python program (default: bench.py, 3 steps)
OUT=${1:-pmc2}; PASSES=${2:-"sq tlb tcc lat"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$OUT/$name -- python3 ${PMC_PROG:-bench.py --steps 2 --warmup 1 --no-cpu-baseline} > gpurun_out/$OUT/$name.json 2> gpurun_out/$OUT/$name.err; }
for p in $PASSES; do case $p in
 sq) run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM ;;
 sq2) run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS ;;
 tlb) run tlb TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum ;;
 tcc) run tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum ;;
 ea) run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum; run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ;;
 tcp) run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum ;;
 lat) run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum ;;
esac; done
ls gpurun_out/$OUT
