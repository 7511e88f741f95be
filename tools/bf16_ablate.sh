# diagnostic: time the bf16 screening kernel with parts removed (results are invalid in these runs)
cd $GRAFT_REPO_ROOT
for a in ${ABLATE_LIST:-0 8 9 13 14 15}; do
  echo "== VDB_BF16_ABLATE=$a"
  VDB_BF16_ABLATE=$a timeout -k 10 120 python tools/kernel_time.py 2>&1 | tail -1
done
