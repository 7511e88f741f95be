# kernel timeline of synchronous C2 steps: per kernel of the chain its duration and the idle gap in front of it
# (rocprofv3 --kernel-trace only; medians over the timed steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --steps 60 --warmup 10 --no-cpu --no-f32-tier --no-gauss --no-pipelined --no-shadow > gpurun_out/tl.log 2>&1
tail -c 200 gpurun_out/tl.log
python - <<'PY'
import csv, glob, statistics as st
f = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)[0]
k = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
starts = [i for i, e in enumerate(k) if 'query_prep' in e[2]]
steps = [k[a:b] for a, b in zip(starts[-50:-1], starts[-49:])]
n = st.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
print(len(steps), 'steps of', n, 'kernels')
prev_end = None
tot_d = tot_g = 0
for j in range(n):
    d = st.median(s[j][1] - s[j][0] for s in steps) / 1e3
    g = st.median((s[j][0] - s[j - 1][1]) for s in steps) / 1e3 if j else float('nan')
    tot_d += d; tot_g += 0 if j == 0 else g
    print(f"{j} gap {g:7.2f} us  dur {d:8.2f} us  {steps[0][j][2][:70]}")
period = st.median(b[0][0] - a[0][0] for a, b in zip(steps, steps[1:])) / 1e3
print(f"sum dur {tot_d:.1f}  in-chain gaps {tot_g:.1f}  period {period:.1f}  between steps {period - tot_d - tot_g:.1f}")
PY
