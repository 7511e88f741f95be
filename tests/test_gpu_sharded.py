"""Row shards + the HIP merge kernel on ONE GPU: W independent GpuFlatIndex shards are searched and
their partial top-k lists merged with vdb_merge_topk_device; the result must equal the unsharded
index and the oracle bit for bit.  (The collective itself is covered on CPU by test_sharded_cpu.py.)"""
import os

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
    from sharded_mirror import merge_topk_torch
    from vectordb_from_scratch_amd.sharded import gpu_local_search, merge_topk_hip, shard_range
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
    from sharded_mirror import merge_topk_torch
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


def test_two_half_search_for_the_single_sync_exchange():
    """vdb_flat_search_batch_device_begin / _finish: the first tier is only enqueued, a device word tells whether the
    host still has work (uncertified queries, errors), finish() then does exactly what the plain call would."""
    vdb = load_package()
    vdb.build()
    rng = np.random.default_rng(77)
    n, d, B, k = 90_000, 40, 21, 10
    dev = torch.device("cuda", 0)

    def run(ix, q):
        q_t = torch.from_numpy(q).to(dev)
        ids = torch.empty((B, k), dtype=torch.int64, device=dev)
        ds = torch.empty((B, k), dtype=torch.float32, device=dev)
        cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        code = torch.full((1,), -7, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ix.search_batch_device_begin(q_t.data_ptr(), B, d, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), code_ptr=code.data_ptr())
        first = int(code.item())                                # ordered behind the search by the begin call itself
        changed = ix.search_batch_device_finish()
        torch.cuda.synchronize()
        return first, changed, ids.cpu().numpy().astype(np.uint64), ds.cpu().numpy(), cnt.cpu().numpy()

    # clean batch: code 0, nothing rewritten, results = the plain call = the oracle
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(0), keep_host_copy=False)
    ix.add_bulk(rows)
    first, changed, gi, gd, gc = run(ix, q)
    assert first == 0 and changed is False
    pi, pdist, pc = ix.search_batch_arrays(q, k)
    assert np.array_equal(gi, pi) and np.array_equal(gd.view(np.uint32), pdist.view(np.uint32)) and np.all(gc == k)
    oi, od = oracle.flat_search(0, rows, q[3], k)
    assert np.array_equal(gi[3], oi) and np.array_equal(gd[3], od)
    # exact ties beyond the re-rank depth (every row 300 times, depth 256): nothing can be certified on the device
    # -> VDB_PENDING_HOST, finish() rewrites
    base = rng.random((300, d), dtype=np.float32)
    rows2 = np.concatenate([base] * 300, 0)
    q2 = np.ascontiguousarray(np.tile(base[:3], (7, 1))[:B])
    ix2 = vdb.GpuFlatIndex(vdb.DistanceMetric(0), keep_host_copy=False)
    ix2.add_bulk(rows2)
    first, changed, gi, gd, gc = run(ix2, q2)
    assert first == 100 and changed is True
    for b in (0, 1, 20):
        oi, od = oracle.flat_search(0, rows2, q2[b], k)
        assert np.array_equal(gi[b], oi) and np.array_equal(gd[b], od)
    # a zero-norm query under Cosine: found on the device -> code 100, finish() raises like the plain call
    ix3 = vdb.GpuFlatIndex(vdb.DistanceMetric(1), keep_host_copy=False)
    ix3.add_bulk(rows)
    qz = q.copy()
    qz[5] = 0.0
    q_t = torch.from_numpy(qz).to(dev)
    ids = torch.empty((B, k), dtype=torch.int64, device=dev); ds = torch.empty((B, k), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev); code = torch.zeros((1,), dtype=torch.int32, device=dev)
    ix3.search_batch_device_begin(q_t.data_ptr(), B, d, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), code_ptr=code.data_ptr())
    assert int(code.item()) == 100
    with pytest.raises(vdb.InvalidVector):
        ix3.search_batch_device_finish()
    assert len(ix3.search(vdb.Vector(q[0]), 3)) == 3            # the handle is unlocked and usable again
    with pytest.raises(vdb.VectorDbError):
        ix3.search_batch_device_finish()                        # no search pending


def test_c_abi_shard_group_single_rank_runs_the_full_rccl_exchange(vdb):
    """include/vdb_shard.h on the one-GPU box: a single-rank RCCL communicator (unique id, ncclCommInitRank, ncclCommCount),
    the local search in two halves writing into the packed buffer, ncclAllGather, the merge kernel, the status reduction.
    Results must equal the plain search and the oracle; a clean batch costs ONE collective, a batch whose first tier
    leaves queries for the host costs TWO; errors come back as the reference's error classes."""
    import torch
    from vectordb_from_scratch_amd.sharded import ShardGroup, group_search
    rng = np.random.default_rng(31)
    n, d, B, k = 90_000, 48, 33, 10
    dev = torch.device("cuda", 0)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((B, d)).astype(np.float32)
    grp = ShardGroup(ShardGroup.unique_id(), 0, 1, device=0)
    assert grp.world() == 1
    for metric in (0, 1, 2):
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows, first_id=1000)                        # global ids of this "shard"
        search = group_search(grp, ix)
        gi, gd, gc = (t.cpu().numpy() for t in search(torch.from_numpy(q).to(dev), k))
        st = grp.last_stats()
        assert st["collectives"] == 1 and st["ranks"] == 1 and st["local_pending"] == 0, st
        pi, pdist, pc = ix.search_batch_arrays(q, k)
        assert np.array_equal(gi.astype(np.uint64), pi) and np.array_equal(gd.view(np.uint32), pdist.view(np.uint32)) and np.all(gc == k)
        for b in (0, 17, B - 1):
            oi, od = oracle.flat_search(metric, rows, q[b], k, ids=np.arange(1000, 1000 + n, dtype=np.uint64))
            assert np.array_equal(gi[b].astype(np.uint64), oi) and np.array_equal(gd[b].view(np.uint32), od.view(np.uint32))
    # the first tier cannot certify (every row 300 times): VDB_PENDING_HOST -> second exchange, results still the oracle's
    base = rng.random((300, d), dtype=np.float32)
    rows2 = np.concatenate([base] * 300, 0)
    q2 = np.ascontiguousarray(np.tile(base[:3], (11, 1))[:B])
    ix2 = vdb.GpuFlatIndex(vdb.DistanceMetric(0), keep_host_copy=False)
    ix2.add_bulk(rows2)
    gi, gd, gc = (t.cpu().numpy() for t in group_search(grp, ix2)(torch.from_numpy(q2).to(dev), k))
    st = grp.last_stats()
    assert st["collectives"] == 2 and st["local_pending"] == 1, st
    for b in (0, 1, 20):
        oi, od = oracle.flat_search(0, rows2, q2[b], k)
        assert np.array_equal(gi[b].astype(np.uint64), oi) and np.array_equal(gd[b], od)
    # errors: a zero-norm query under Cosine (found on the device, reported by finish) and a dimension mismatch (refused by begin)
    ix3 = vdb.GpuFlatIndex(vdb.DistanceMetric(1), keep_host_copy=False)
    ix3.add_bulk(rows)
    qz = q.copy()
    qz[5] = 0.0
    with pytest.raises(vdb.InvalidVector):
        group_search(grp, ix3)(torch.from_numpy(qz).to(dev), k)
    with pytest.raises(vdb.DimensionMismatch):
        group_search(grp, ix3)(torch.from_numpy(np.ascontiguousarray(q[:, :20])).to(dev), k)
    gi, _, gc = group_search(grp, ix3)(torch.from_numpy(q).to(dev), k)      # group and handle stay usable
    assert int(gc.min()) == k
    # an empty shard contributes nothing; k = 0 and nq = 0 involve no collective
    ix4 = vdb.GpuFlatIndex(vdb.DistanceMetric(0), keep_host_copy=False)
    _, _, gc = group_search(grp, ix4)(torch.from_numpy(q).to(dev), k)
    assert int(gc.max()) == 0
    # world = 1 WITHOUT an id: no RCCL at all, the plain local search
    g1 = ShardGroup(None, 0, 1, device=0)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(2), keep_host_copy=False)
    ix.add_bulk(rows)
    gi, gd, gc = (t.cpu().numpy() for t in group_search(g1, ix)(torch.from_numpy(q).to(dev), k))
    assert g1.last_stats()["collectives"] == 0
    pi, pdist, _ = ix.search_batch_arrays(q, k)
    assert np.array_equal(gi.astype(np.uint64), pi) and np.array_equal(gd.view(np.uint32), pdist.view(np.uint32))


def test_torch_distributed_fallback_exchange_single_rank(vdb):
    """bench.py's fallback when the library's own RCCL group cannot be created on every rank: the same exchange with
    torch.distributed as the transport (sharded.torch_group_search).  One rank here: all-gathers over the "nccl" backend (RCCL),
    the HIP merge, results equal to the plain search; a failing local search raises after the collectives."""
    import torch
    import torch.distributed as dist
    from vectordb_from_scratch_amd.sharded import torch_group_search
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(41)
        n, d, B, k = 70_000, 64, 19, 7
        rows = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((B, d)).astype(np.float32)
        for metric in (0, 1, 2):
            ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
            ix.add_bulk(rows, first_id=500)
            gi, gd, gc = (t.cpu().numpy() for t in torch_group_search(ix)(torch.from_numpy(q).cuda(), k))
            pi, pdist, pc = ix.search_batch_arrays(q, k)
            assert np.array_equal(gi.astype(np.uint64), pi) and np.array_equal(gd.view(np.uint32), pdist.view(np.uint32)) and np.all(gc == k)
        ixc = vdb.GpuFlatIndex(vdb.DistanceMetric(1), keep_host_copy=False)
        ixc.add_bulk(rows)
        qz = q.copy()
        qz[3] = 0.0
        with pytest.raises(vdb.InvalidVector):
            torch_group_search(ixc)(torch.from_numpy(qz).cuda(), k)
        gi, _, gc = torch_group_search(ixc)(torch.from_numpy(q).cuda(), k)       # usable afterwards
        assert int(gc.min()) == k
    finally:
        if own:
            dist.destroy_process_group()
