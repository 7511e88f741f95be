# diagnostic: depth the adaptive re-rank ends at, and the step / re-rank time for several first-round depths
# the knobs below exist only in the diagnostics build (make -C vectordb-from-scratch_amd/csrc diag)
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VDB_RR_DEPTH=1 timeout -k 10 200 python tools/kernel_time.py 2>&1 | grep "re-rank depth" | tail -2
for f in ${KP_FIRST_LIST:-32 48}; do
  echo "== VDB_KP_FIRST=$f"
  rm -rf gpurun_out/profR
  VDB_KP_FIRST=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profR -- python tools/kernel_time.py > gpurun_out/profR.log 2>&1
  tail -1 gpurun_out/profR.log
  grep "rerank_kernel\|select_kernel" gpurun_out/profR/*/*kernel_stats.csv | cut -d, -f1-4
done
