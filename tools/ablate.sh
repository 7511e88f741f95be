# diagnostic: time the fused kernel with phases switched off (results are wrong in these modes)
# the knobs below exist only in the diagnostics build (make -C vectordb-from-scratch_amd/csrc diag)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for a in ${ABLATE_MODES:-0 1 6}; do echo -n "ablate=$a kernel_ms,step_ms: "; VDB_FUSED_ABLATE=$a timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], d['ms_per_step'])"; done
