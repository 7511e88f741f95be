cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the knobs below exist only in the diagnostics build (make -C vectordb-from-scratch_amd/csrc diag)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for b in 0 1; do
  echo "== VDB_SAMPLE_BLOCK=$b"
  rm -rf gpurun_out/profB
  if [ $b = 1 ]; then export VDB_SAMPLE_BLOCK=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profB -- python tools/kernel_time.py > gpurun_out/profB.log 2>&1
  tail -1 gpurun_out/profB.log
  grep "fused_bf16_kernel<true>\|select_kernel\|rerank_kernel" gpurun_out/profB/*/*kernel_stats.csv | cut -d, -f1-4
done
