# what one rank of a strong-scaling run does per step: the C2 workload at the shard sizes of 2 / 4 / 8 GPUs (500k / 250k / 125k rows),
# per-kernel averages under rocprofv3 and the unprofiled step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for R in 500000 250000 125000; do
  python bench.py --rows $R --steps 100 --warmup 10 --lean > gpurun_out/shard_$R.json 2> gpurun_out/shard_$R.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profR$R -- python bench.py --rows $R --steps 60 --warmup 10 --lean > gpurun_out/profR$R.log 2>&1
  python - <<PY
import json,glob,csv
l=json.loads(open("gpurun_out/shard_$R.json").read().strip().splitlines()[-1])
print("rows $R: ms_per_step", l["ms_per_step"], "value", l["value"], "kernel_ms", l["roofline"].get("kernel_ms"))
f=glob.glob("gpurun_out/profR$R/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:9]:
    print("   %-60s calls %5s avg %8.1f us min %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
