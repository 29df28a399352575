# MFMA pipe utilisation of the dense kernels (north_star: "MFMA utilisation against chip peak"): busy cycles of the
# matrix pipe vs. the kernel's cycles, per kernel.  One pass, SQ counters only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mfma/a.json 2> gpurun_out/pmc_mfma/a.err
ls gpurun_out/pmc_mfma/a/*/ | head
