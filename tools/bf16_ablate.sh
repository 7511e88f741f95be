# diagnostic: time the bf16 screening kernel with parts removed (results are invalid in these runs)
# the knobs below exist only in the diagnostics build (make -C vectordb-from-scratch_amd/csrc diag)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
# bits: 2 no row DMA, 4 no query DMA, 8 no epilogue, 16 thresholds = -inf (nothing passes: the cost of the append path);
# bit 1 (no LDS reads + MFMAs) exists in the unpipelined kernel only: VDB_FUSED_PIPE=0 ABLATE_LIST="0 8 9 13 14 15"
cd $GRAFT_REPO_ROOT
for a in ${ABLATE_LIST:-0 16 8}; do
  echo "== VDB_BF16_ABLATE=$a"
  VDB_BF16_ABLATE=$a timeout -k 10 120 python tools/kernel_time.py 2>&1 | tail -1
done
