# A/B of the threshold rank kt (diagnostics build): expected keys per query = kt * N / S
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null && export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so
for kt in 0 16 24 0 16; do
  echo "== VDB_KT16=$kt"
  if [ $kt = 0 ]; then timeout -k 10 200 python bench.py --no-gauss --no-pipelined --no-shadow --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
  else VDB_KT16=$kt timeout -k 10 200 python bench.py --no-gauss --no-pipelined --no-shadow --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"; fi
done
