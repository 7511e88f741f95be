cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcA3 -- python bench.py --config c3 --steps 3 --warmup 1 --lean > gpurun_out/pmcA3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcB3 -- python bench.py --config c3 --steps 3 --warmup 1 --lean > gpurun_out/pmcB3.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("pmcA3", "pmcB3"):
    f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fused_bf16w" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: round(sum(v) / len(v), 1) for k, v in acc.items()}, "launches", max(len(v) for v in acc.values()) if acc else 0)
PY
