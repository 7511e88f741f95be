# API + kernel timeline of search steps (rocprofv3 hip-trace + kernel-trace, no counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace1
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace1 -- python tools/kernel_time.py > gpurun_out/trace1.log 2>&1
tail -1 gpurun_out/trace1.log
ls gpurun_out/trace1/*/
python - <<'PY'
import csv, glob
d = glob.glob('gpurun_out/trace1/*/')[0]
k = list(csv.DictReader(open(glob.glob(d + '*kernel_trace.csv')[0])))
h = list(csv.DictReader(open(glob.glob(d + '*hip_api_trace.csv')[0])))
m = glob.glob(d + '*memory_copy_trace.csv')
mc = list(csv.DictReader(open(m[0]))) if m else []
ev = []
for r in k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:50]))
for r in h: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'A ' + r['Function']))
for r in mc: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M ' + r.get('Direction', '')))
ev.sort()
# find the 10th fused_bf16<false> kernel and print events from the preceding query_prep API call to the next one
idx = [i for i, e in enumerate(ev) if e[2].startswith('K void vdb::fused_bf16_kernel<false>')]
i = idx[10]
j0 = i
while j0 > 0 and 'query_prep' not in ev[j0][2]: j0 -= 1
j0 -= 12
t0 = ev[j0][0]
for e in ev[j0:i + 40]:
    print(f"{e[0]-t0:9d} {e[1]-t0:9d} {e[1]-e[0]:8d}  {e[2]}")
PY
