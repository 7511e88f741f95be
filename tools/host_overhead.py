#!/usr/bin/env python3
"""Where the time of one search step goes on the host: the C call (vdb_flat_last_stats host clocks) against the Python
step period of bench.py's call chain (ShardedSearcher -> gpu_local_search -> ctypes)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package, gen_chunk, gen_queries
n, dim, B, k = int(os.environ.get("KT_ROWS", 1000000)), 768, 256, 10
vdb = load_package(); vdb.build()
from vectordb_from_scratch_amd.sharded import ShardedSearcher, gpu_local_search
dev = torch.device("cuda", 0)
ix = vdb.GpuFlatIndex(vdb.DistanceMetric(1), device=0, keep_host_copy=False)
ix.reserve(n, dim)
for c in range((n + 124999) // 125000):
    m = min(125000, n - c * 125000)
    blk = gen_chunk(c, m, dim, dev); torch.cuda.synchronize()
    ix.add_bulk_device(blk.data_ptr(), m, dim, first_id=c * 125000); del blk
ix.flush()
q = gen_queries(B, dim, dev)
s = ShardedSearcher(gpu_local_search(ix, reuse_outputs=True), rank=0, world=1)
for _ in range(5): s.search_batch(q, k)
torch.cuda.synchronize()
N = 200
tot = []
t0 = time.perf_counter()
for _ in range(N):
    s.search_batch(q, k); tot.append(ix.last_stats()["host_total_ns"])
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / N * 1e6
ls = time.perf_counter()
for _ in range(N): ix.last_stats()
ls = (time.perf_counter() - ls) / N * 1e6
print(f"step period {per - ls:.1f} us (last_stats {ls:.1f} us excluded); inside the C call {np.mean(tot) / 1e3:.1f} us; outside it {per - ls - np.mean(tot) / 1e3:.1f} us")
