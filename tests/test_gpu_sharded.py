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


def test_packed_exchange_buffer_merge():
    """vdb_merge_topk_packed_device reads the all-gathered buffer of sharded.py in place and reduces the status words."""
    import ctypes
    vdb = load_package()
    vdb.build()
    from vectordb_from_scratch_amd import _ffi
    from vectordb_from_scratch_amd.sharded import merge_topk_torch
    W, B, k = 3, 17, 10
    g = torch.Generator().manual_seed(5)
    words = B * (3 * k + 1) + 1
    words += words & 1
    dev = torch.device("cuda", 0)
    ids = torch.randint(0, 1000, (W, B, k), generator=g, dtype=torch.int64)
    dists = torch.rand((W, B, k), generator=g).sort(dim=2).values
    dists[1, :, 3] = dists[0, :, 3]                                   # cross-part distance ties -> decided by id
    counts = torch.randint(0, k + 1, (W, B), generator=g, dtype=torch.int32)
    packed = torch.zeros((W, words), dtype=torch.int32)
    for w in range(W):
        packed[w, :2 * B * k] = ids[w].contiguous().view(torch.int32).view(-1)
        packed[w, 2 * B * k:3 * B * k] = dists[w].contiguous().view(torch.int32).view(-1)
        packed[w, 3 * B * k:3 * B * k + B] = counts[w]
        packed[w, 3 * B * k + B] = w                                  # status words 0, 1, 2 -> max 2
    pd = packed.to(dev)
    out_i = torch.empty((B, k), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, k), dtype=torch.float32, device=dev)
    out_c = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    rc = _ffi.lib().vdb_merge_topk_packed_device(0, ctypes.c_void_p(pd.data_ptr()), W, words, B, k,
                                                 ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(out_d.data_ptr()),
                                                 ctypes.c_void_p(out_c.data_ptr()), ctypes.c_void_p(out_c.data_ptr() + 4 * B), None)
    assert rc == 0
    torch.cuda.synchronize()
    ti, td, tc = merge_topk_torch(ids, dists, counts, k)
    n = tc.tolist()
    assert out_c[:B].cpu().tolist() == n and int(out_c[B]) == 2
    for b in range(B):
        assert out_i[b, :n[b]].cpu().tolist() == ti[b, :n[b]].tolist()
        assert torch.equal(out_d[b, :n[b]].cpu(), td[b, :n[b]])
