# collects the judged evidence for profiles/: bench line, rocprof kernel stats (>= 50 timed steps, so that the AVERAGE -- not
# the minimum -- supports the quoted fraction), PMC passes (each in its own run), and the same for config 3's shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profS -- python bench.py --steps 60 --warmup 10 --lean > gpurun_out/profS.log 2>&1
bash tools/pmc_fused.sh > gpurun_out/pmc_run.log 2>&1
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
# config 3's shape (one 1.25M-row shard, B = 1024, k = 100): kernel stats + the FETCH_SIZE / WRITE_SIZE passes of the wide kernel
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profS3 -- python bench.py --config c3 --steps 30 --warmup 5 --lean > gpurun_out/profS3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF3 -- python bench.py --config c3 --steps 3 --warmup 1 --lean > gpurun_out/pmcF3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW3 -- python bench.py --config c3 --steps 3 --warmup 1 --lean > gpurun_out/pmcW3.log 2>&1
python bench.py --config c1 --steps 200 --warmup 20 > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err
python bench.py --config c4 --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err
ls gpurun_out/profS/*/ gpurun_out/profS3/*/ gpurun_out/pmcF3/*/ | head -30
