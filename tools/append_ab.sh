# A/B of the filter kernel's stores (diagnostics build, same device, medians of KT_ITERS launches):
#   0 as shipped   32 hits counted, keys not written   16 nothing passes   2048 sub-pool counts not written
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for a in ${AB_LIST:-0 2048 32 2080 16}; do
  echo "== VDB_BF16_ABLATE=$a"
  KT_ITERS=${KT_ITERS:-60} VDB_BF16_ABLATE=$a timeout -k 10 120 python tools/kernel_time.py 2>&1 | tail -1
done
