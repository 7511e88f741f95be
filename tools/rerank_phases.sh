# diagnostic: where a re-rank workgroup spends its time (s_memrealtime stamps; diagnostics build, VDB_RR_DEPTH)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for f in ${KP_FIRST_LIST:-48}; do
  echo "== VDB_KP_FIRST=$f"
  VDB_KP_FIRST=$f VDB_RR_DEPTH=1 timeout -k 10 200 python tools/kernel_time.py 2>&1 | grep "re-rank" | tail -8
done
