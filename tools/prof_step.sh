# per-kernel times of one search step (rocprofv3 kernel trace) at 1M x 768, batch 256
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profK
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/profK -- python tools/kernel_time.py > gpurun_out/profK.log 2>&1
tail -1 gpurun_out/profK.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/profK/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if 'fused_bf16_kernel<false>' in r['Kernel_Name'] or 'fused_score_filter' in r['Kernel_Name']]
i = idx[len(idx) // 2]
j0 = i
while j0 > 0 and 'query_prep' not in rows[j0]['Kernel_Name']: j0 -= 1
t0 = int(rows[j0]['Start_Timestamp'])
for r in rows[j0:i + 4]:
    print(f"{r['Kernel_Name'][:64]:64s} start {int(r['Start_Timestamp'])-t0:8d} dur {int(r['End_Timestamp'])-int(r['Start_Timestamp']):8d}")
PY
