#!/usr/bin/env python3
"""Cluster-ordered data (rows stored cluster by cluster, queries next to stored rows): the workload on which the screening
tier exhausts its re-rank depth.  Times a batch with and without the re-threshold pass (vdb_flat_set_tiers(VDB_TIERS_NO_RETHRESHOLD))."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package
import oracle
vdb = load_package(); vdb.build()
rng = np.random.default_rng(46)
n, d, nq, k = int(os.environ.get("CB_ROWS", 400000)), 768, 256, 10
c = rng.standard_normal((n // 500, d)).astype(np.float32) * 3.0
lab = np.sort(rng.integers(0, c.shape[0], n))
rows = (c[lab] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
q = (rows[rng.integers(0, n, nq)] + 1e-3 * rng.standard_normal((nq, d))).astype(np.float32)
ix = vdb.GpuFlatIndex(vdb.DistanceMetric(1), keep_host_copy=False)
ix.add_bulk(rows)
r = ix.search_batch_arrays(q, k)
t0 = time.perf_counter()
for _ in range(5): r = ix.search_batch_arrays(q, k)
ms = (time.perf_counter() - t0) * 200
st = ix.last_stats()
oi, od = oracle.flat_search(1, rows, q[7], k)
ok = np.array_equal(oi, r[0][7]) and np.array_equal(od.view(np.uint32), r[1][7].view(np.uint32))
print(f"n={n} cosine clustered, batch {nq}: {ms:.2f} ms per batch = {nq / ms * 1e3:.0f} queries/s; rethreshold {st['rethreshold_queries']} f32-tier {st['f32_tier_queries']} exact {st['exact_queries']}; oracle parity {ok}")
