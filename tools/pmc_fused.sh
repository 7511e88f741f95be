# diagnostic: PMC passes over the fused kernel (separate passes; no tracing domains beside kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcA -- python bench.py --steps 3 --warmup 1 --lean > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcB -- python bench.py --steps 3 --warmup 1 --lean > gpurun_out/pmcB.log 2>&1
ls gpurun_out/pmcA/*/ gpurun_out/pmcB/*/
