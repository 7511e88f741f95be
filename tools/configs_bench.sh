# measurement of the other BASELINE configurations on one GPU (the headline line is plain `python bench.py`)
run() { echo -n "$1: "; shift; timeout -k 10 280 python bench.py --no-f32-tier --no-gauss "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; s=d['path_stats']; p=d.get('pipelined_two_in_flight') or {}; print('QPS', d['value'], 'step_ms', d['ms_per_step'], 'in-flight step_ms', p.get('ms_per_step'), 'kernel_ms', r['kernel_ms'], r['bound'], r['achieved'], r['unit'], 'recall', d['recall_at_k'], 'f32q', s['f32_tier_queries'], 'rethr', s['rethreshold_queries'], 'exact', s['exact_queries'], 'ovf', s['pool_overflows'], (d['cpu_baseline'] or {}).get('ids_and_distances_bit_identical'))"; }
run "C2 cosine 1Mx768 B256 k10" --steps 100 --warmup 10 --cpu-seconds 6
run "C2 euclid" --metric 0 --steps 60 --warmup 10 --cpu-seconds 3
run "C2 dot" --metric 2 --steps 60 --warmup 10 --cpu-seconds 3
run "C3 shard: dot 1.25Mx768 B1024 k100" --config c3 --steps 20 --warmup 4 --cpu-seconds 4
run "C4 euclid 1Mx1536 B256 k10, eq filter on string metadata" --config c4 --steps 40 --warmup 8 --cpu-seconds 6
run "k=30 cosine (3k over-fetch)" --k 30 --steps 40 --warmup 8 --cpu-seconds 3
run "B=1 cosine" --batch 1 --steps 50 --warmup 5 --cpu-seconds 2
run "B=32 cosine" --batch 32 --steps 50 --warmup 5 --cpu-seconds 2
run "B=128 cosine" --batch 128 --steps 30 --warmup 5 --no-cpu
run "C2 cosine, f32 MFMA tier only" --screen 0 --steps 20 --warmup 3 --no-cpu
