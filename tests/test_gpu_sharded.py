"""Row shards + the HIP merge kernel on ONE GPU: W independent GpuFlatIndex shards are searched and
their partial top-k lists merged with vdb_merge_topk_device; the result must equal the unsharded
index and the oracle bit for bit.  (The collective itself is covered on CPU by test_sharded_cpu.py.)"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("world", [2, 4])
def test_shards_merge_equals_unsharded(metric, world):
    vdb = load_package()
    vdb.build()
    from vectordb_from_scratch_amd.sharded import gpu_local_search, merge_topk_hip, merge_topk_torch, shard_range
    rng = np.random.default_rng(100 + metric)
    n, d, B, k = 50_000, 48, 37, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[123] = rows[40_000]                                   # exact tie across shards
    queries = rng.standard_normal((B, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    q_t = torch.from_numpy(queries).to(dev)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows[lo:hi], first_id=lo)
        parts.append(gpu_local_search(ix)(q_t, k))
    g_i = torch.stack([p[0] for p in parts])
    g_d = torch.stack([p[1] for p in parts])
    g_c = torch.stack([p[2] for p in parts])
    mi, md, mc = merge_topk_hip(g_i, g_d, g_c, k)
    ti, td, tc = merge_topk_torch(g_i, g_d, g_c, k)
    torch.cuda.synchronize()
    assert torch.equal(mi, ti) and torch.equal(md, td) and torch.equal(mc, tc)
    full = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    full.add_bulk(rows)
    fi, fd, fc = full.search_batch_arrays(queries, k)
    assert np.array_equal(mi.cpu().numpy().astype(np.uint64), fi) and np.array_equal(md.cpu().numpy(), fd)
    for b in range(0, B, 5):
        oi, od = oracle.flat_search(metric, rows, queries[b], k)
        assert np.array_equal(mi[b].cpu().numpy().astype(np.uint64), oi) and np.array_equal(md[b].cpu().numpy(), od)
