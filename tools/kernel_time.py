#!/usr/bin/env python3
"""Kernel time of the dominant kernel (HIP events, vdb_flat_set_profile) and step time at 1M x 768, batch 256."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package, gen_chunk, gen_queries
n, dim, B, k = int(os.environ.get("KT_ROWS", 1000000)), int(os.environ.get("KT_DIM", 768)), int(os.environ.get("KT_BATCH", 256)), int(os.environ.get("KT_K", 10))
metric = int(os.environ.get("KT_METRIC", 1))
vdb = load_package(); vdb.build()
dev = torch.device("cuda", 0)
ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), device=0, keep_host_copy=False)
ix.reserve(n, dim)
for c in range((n + 124999) // 125000):
    m = min(125000, n - c * 125000)
    blk = gen_chunk(c, m, dim, dev); torch.cuda.synchronize()
    ix.add_bulk_device(blk.data_ptr(), m, dim, first_id=c * 125000); del blk
ix.flush()
ix.set_screen(int(os.environ.get("KT_SCREEN", 1)))
if os.environ.get("KT_SHADOW"): ix.set_shadow(True)
q = gen_queries(B, dim, dev)
ids = torch.empty((B, k), dtype=torch.int64, device=dev); ds = torch.empty((B, k), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
def step(): ix.search_batch_device(q.data_ptr(), B, dim, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr())
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
hs = []
for _ in range(20):
    step(); st_ = ix.last_stats(); hs.append((st_["host_enqueued_ns"], st_["host_flags_ns"], st_["host_total_ns"]))
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
print("host us (enqueued, flags on host, total):", np.mean(hs, axis=0) / 1e3)
ix.set_profile(True); ks = []
for _ in range(int(os.environ.get("KT_ITERS", 10))): step(); ks.append(ix.last_stats()["fused_kernel_ns"] / 1e6)
st = ix.last_stats()
print(f"shadow_rows {st['shadow_rows']}  step {ms:.3f} ms  kernel {np.mean(ks):.4f} ms (median {np.median(ks):.4f}, min {min(ks):.4f})  {4.0*n*dim/np.mean(ks)/1e9:.2f} TB/s  uncert {st['uncertified']} f32q {st['f32_tier_queries']} ovf {st['pool_overflows']}")
