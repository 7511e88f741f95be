# diagnostic A/B: two 4-wave workgroups per CU (default) vs one 8-wave workgroup per CU
# the knobs below exist only in the diagnostics build (make -C vectordb-from-scratch_amd/csrc diag)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for rep in 1 2; do
for v in "" 1; do echo -n "SHAPE8=${v:-0} "; env ${v:+VDB_FUSED_SHAPE8=1} timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('kernel_ms', r['kernel_ms'], 'TF', r['achieved'], 'step_ms', d['ms_per_step'], 'exact', d['path_stats']['exact_queries'])"; done; done
