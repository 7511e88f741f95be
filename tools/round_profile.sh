# collects the judged evidence for profiles/: bench line, rocprof kernel stats, PMC passes (each in its own run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profS -- python bench.py --steps 10 --warmup 2 --no-cpu --no-f32-tier --no-gauss --no-pipelined --no-shadow > gpurun_out/profS.log 2>&1
bash tools/pmc_fused.sh > gpurun_out/pmc_run.log 2>&1
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu 2>/dev/null | tail -1 | cut -c1-200
