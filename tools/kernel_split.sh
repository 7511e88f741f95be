# per-kernel split of a SYNCHRONOUS step at C2 (rocprofv3 --kernel-trace --stats; no pipelined / f32 / gauss side runs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profS -- python bench.py --steps 30 --warmup 5 --no-cpu --no-f32-tier --no-gauss --no-pipelined --no-shadow > gpurun_out/profS.log 2>&1
tail -c 300 gpurun_out/profS.log
