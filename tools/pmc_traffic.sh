# diagnostic: HBM traffic of the fused kernel (FETCH_SIZE and WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- python bench.py --steps 3 --warmup 1 --lean > gpurun_out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- python bench.py --steps 3 --warmup 1 --lean > gpurun_out/pmcW.log 2>&1
ls gpurun_out/pmcF/*/ gpurun_out/pmcW/*/
