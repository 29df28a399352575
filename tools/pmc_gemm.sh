cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc3
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc3/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc3/$name.json 2> gpurun_out/pmc3/$name.err; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
run sq2 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
