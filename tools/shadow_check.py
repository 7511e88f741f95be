#!/usr/bin/env python3
"""bf16 shadow rows against the f32-row screening pass: identical ids / distance bits / counts and identical raw screening
scores, for the three metrics, ragged row counts and a row pitch the shadow kernel cannot take (falls back)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package
vdb = load_package(); vdb.build()
rng = np.random.default_rng(3)
bad = 0
for metric in (1, 0, 2):
    for n, dim, B, k in ((70001, 128, 256, 10), (150000, 768, 100, 10), (66000, 64, 7, 50), (90000, 200, 256, 10), (70000, 96, 33, 5)):
        rows = rng.random((n, dim), dtype=np.float32) - (0.3 if metric == 2 else 0.0)
        q = rng.random((B, dim), dtype=np.float32)
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows)
        i0, d0, c0 = ix.search_batch_arrays(q, k)
        s0 = ix.last_stats()
        ix.set_shadow(True)
        i1, d1, c1 = ix.search_batch_arrays(q, k)
        s1 = ix.last_stats()
        ok = np.array_equal(i0, i1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(c0, c1)
        same_stats = all(s0[key] == s1[key] for key in ("uncertified", "pool_overflows", "f32_tier_queries", "exact_queries", "rethreshold_queries"))
        print(f"metric {metric} n {n} dim {dim} B {B} k {k}: shadow used {s1['shadow_rows']} (before {s0['shadow_rows']})  results equal {ok}  tier stats equal {same_stats}  uncert {s1['uncertified']}", flush=True)
        bad += (not ok) or (not same_stats) or s0['shadow_rows'] != 0 or s1['shadow_rows'] != (1 if ((dim + 31) // 32 * 32) % 64 == 0 else 0)
        del ix
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
