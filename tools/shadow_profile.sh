# evidence for the opt-in bf16 shadow rows (vdb_flat_set_shadow): kernel stats and PMC passes of one search step at
# 1M x 768, batch 256 (tools/kernel_time.py with VDB_SHADOW=1); every PMC pass in its own run, kernel-trace only beside it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VDB_SHADOW=1
rm -rf gpurun_out/shS gpurun_out/shA gpurun_out/shF gpurun_out/shW
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shS -- python tools/kernel_time.py > gpurun_out/shS.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/shA -- python tools/kernel_time.py > gpurun_out/shA.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/shF -- python tools/kernel_time.py > gpurun_out/shF.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/shW -- python tools/kernel_time.py > gpurun_out/shW.log 2>&1
tail -1 gpurun_out/shS.log
