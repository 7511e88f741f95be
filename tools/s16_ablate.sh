# where the bf16-shadow filter pass spends its time (diagnostics build): VDB_BF16_ABLATE bits 2 no row DMA, 4 no query DMA,
# 8 no epilogue, 16 nothing passes the filter
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for a in ${S16_ABLATE_LIST:-0 6 2 4 8 16 14}; do
  echo "== VDB_BF16_ABLATE=$a"
  KT_ITERS=${KT_ITERS:-40} KT_SHADOW=1 VDB_BF16_ABLATE=$a timeout -k 10 120 python tools/kernel_time.py 2>&1 | tail -1
done
