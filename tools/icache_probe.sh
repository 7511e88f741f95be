# diagnostic: instruction-cache behaviour of the screening kernel (are the out-of-line append blocks of the epilogue missing?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INSTS_VALU\b\|SQ_INSTS_SALU\b\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES" | sort -u | tr '\n' ' '; echo
make -C vectordb-from-scratch_amd/csrc -j8 diag >/dev/null
for a in 0 16; do
  export VDB_LIB=$PWD/vectordb-from-scratch_amd/libvdbflat_diag.so VDB_BF16_ABLATE=$a
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pmcI$a -- python tools/kernel_time.py > gpurun_out/pmcI$a.log 2>&1
  python - <<PY
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/pmcI$a/*/*counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "fused_bf16p" in r["Kernel_Name"]: agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
last = sorted(agg, key=int)[-1]
print("ablate=$a", {k: int(v) for k, v in sorted(agg[last].items())})
PY
done
