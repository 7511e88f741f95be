# diagnostic: where the append path of the screening kernel's epilogue spends its ~40 us (diagnostics build only)
#   0 full kernel, 16 thresholds = -inf (nothing passes), 32 hits counted but not stored, 64 branch taken but append skipped
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for a in 0 16 32 64 0 16; do
  echo "== VDB_BF16_ABLATE=$a"
  VDB_BF16_ABLATE=$a timeout -k 10 120 python tools/kernel_time.py 2>&1 | tail -1
done
