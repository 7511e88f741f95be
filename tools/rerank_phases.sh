# diagnostic: where a re-rank workgroup spends its time (s_memrealtime stamps; diagnostics build, VDB_RR_DEPTH) and the depth
# each query's re-rank ended at.  KT_METRIC / KT_K / KT_ROWS / KT_DIM choose the shape (default config 2).
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for f in ${KP_FIRST_LIST:-0}; do
  echo "== VDB_KP_FIRST=$f"
  VDB_KP_FIRST=$f VDB_RR_DEPTH=1 timeout -k 10 200 python tools/kernel_time.py 2>&1 | grep "re-rank" | tail -8
done
