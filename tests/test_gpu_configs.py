"""BASELINE.json configs 3, 4 and 5 AT THEIR OWN SHAPE inside `pytest -m gpu` (VERDICT r1 "configs_untested"):

  C3  one shard of the 10M x 768 dot-product job: 1.25M x 768, batch 1024 (two passes of the 512-query kernel), k = 100, through the
      C-ABI shard group (single-rank RCCL communicator: local search in two halves, all-gather, merge);
  C4  1M x 1536 Euclidean with the 25 % eq-filter mask that VectorStore.compile_filter builds from real string metadata;
  C5  the HNSW index at 768 dimensions, m = 16, ef_search = 200, against the CPU restatement of the reference's HNSW.

Each is checked through size-independent properties, two whole queries against the CPU oracle (seconds of CPU each) and,
for C4, the prefix property that ties the device pre-filter to the reference's post-filter (src/storage.rs:268-287)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vdb():
    v = load_package()
    v.build()
    return v


def test_c1_flat_10k_x_128_euclid_single_query_at_its_own_shape(vdb):
    """BASELINE configs[0] = benches/search_bench.rs:18-33: 10,000 x 128 uniform[0,1) rows, Euclidean, k = 10, ONE query
    [0.5; 128] -- the literal FlatIndex::search drop-in (src/flat_index.rs:52-65), one vdb_flat_search call.  10k rows take the
    dense-scores path (indexes up to 16384 rows: every row scored by the f32-input MFMA kernel, no filter pass), the only size
    of BASELINE that does -- and, for batches of up to 8 queries, the DIRECT path: one exact-scan kernel + one select/emit kernel
    through mapped host memory (the latency of a call is the measure here, bench.py --config c1).  Whole result against the
    oracle, bit for bit, through both; k = 10000 (the full sort the reference performs) through the exact-scan path as well."""
    import ctypes
    n, d, k = 10_000, 128, 10
    rows = np.random.default_rng(0).random((n, d), dtype=np.float32)
    q = np.full((d,), 0.5, dtype=np.float32)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    for i in range(0, 64):                                          # Index::add one row at a time, as the bench's setup does...
        ix.add(i, vdb.Vector(rows[i]))
    ix.add_bulk(rows[64:], first_id=64)                             # ... the rest in one call
    L = vdb._ffi.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    out_i, out_d, out_c = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32), ctypes.c_size_t()
    rc = L.vdb_flat_search(ix._h, q.ctypes.data_as(fp), d, k, out_i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), out_d.ctypes.data_as(fp),
                           ctypes.byref(out_c))
    assert rc == 0, vdb._ffi.last_error()
    oi, od = oracle.flat_search(0, rows, q, k)
    assert out_c.value == k and np.array_equal(out_i, oi) and np.array_equal(out_d.view(np.uint32), od.view(np.uint32))
    st = ix.last_stats()
    assert st["exact_queries"] == 1 and st["mfma_queries"] == 0 and st["rows_scanned"] == 0, st    # the direct path: an exact scan, two kernels
    # the same call through the tiered pipeline (dense f32-MFMA scores of every row, certified re-rank): same bits
    ix.set_tiers(ix.TIERS_NO_DIRECT)
    out2_i, out2_d, out2_c = np.zeros(k, dtype=np.uint64), np.zeros(k, dtype=np.float32), ctypes.c_size_t()
    rc = L.vdb_flat_search(ix._h, q.ctypes.data_as(fp), d, k, out2_i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), out2_d.ctypes.data_as(fp),
                           ctypes.byref(out2_c))
    assert rc == 0 and out2_c.value == k and np.array_equal(out2_i, oi) and np.array_equal(out2_d.view(np.uint32), od.view(np.uint32))
    st = ix.last_stats()
    assert st["bf16_screen"] == 0 and st["sample_rows"] == n and st["rows_scanned"] == 0 and st["exact_queries"] == 0, st    # dense scores of every row, certified
    ix.set_tiers(0)
    res = ix.search(vdb.Vector(q), k)                               # the trait method of the Python mirror: same call underneath
    assert [r[0] for r in res] == list(oi) and np.array_equal(np.array([r[1] for r in res], dtype=np.float32).view(np.uint32), od.view(np.uint32))
    # self-query: distance exactly 0 first (distance.rs:89-93)
    r0 = ix.search(vdb.Vector(rows[4321]), 1)
    assert r0[0][0] == 4321 and float(r0[0][1]) == 0.0
    # the reference sorts ALL rows (flat_index.rs:62): k = len gives the whole ordering
    gi, gd, gc = ix.search_batch_arrays(q[None, :], n)
    fi, fd = oracle.flat_search(0, rows, q, n)
    assert gc[0] == n and np.array_equal(gi[0], fi) and np.array_equal(gd[0].view(np.uint32), fd.view(np.uint32))
    # the sharded handle at this shape: 3 shards of 3334 / 3333 / 3333 rows, same answer
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 0, 0], keep_host_copy=False)
    sh.add_bulk(rows)
    si, sd, sc = sh.search_batch_arrays(q[None, :], k)
    assert sc[0] == k and np.array_equal(si[0], oi) and np.array_equal(sd[0].view(np.uint32), od.view(np.uint32))


def test_c3_one_shard_of_the_10m_job_through_the_shard_group(vdb):
    from vectordb_from_scratch_amd.sharded import ShardGroup, group_search
    n, d, B, k = 1_250_000, 768, 1024, 100
    rank, world = 3, 8                                            # this shard's place in the 10M-row job: global ids 3.75M ..
    first_id = rank * n
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1003)
    rows = torch.rand((n, d), generator=g, device=dev, dtype=torch.float32)
    g.manual_seed(4)
    queries = torch.rand((B, d), generator=g, device=dev, dtype=torch.float32)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.DotProduct, keep_host_copy=False)
    ix.add_bulk_device(rows.data_ptr(), n, d, first_id=first_id)
    ix.flush()
    grp = ShardGroup(ShardGroup.unique_id(), 0, 1, device=0)     # the full exchange path on the one GPU of this box
    search = group_search(grp, ix)
    ids, dists, counts = search(queries, k)
    torch.cuda.synchronize()
    st, gs = ix.last_stats(), grp.last_stats()
    assert gs["collectives"] == 1 and gs["ranks"] == 1, gs
    assert st["bf16_screen"] == 1 and st["kprime"] == 512 and st["rows_scanned"] == 2 * n, st      # two passes of 512 queries: the shard is read twice per batch
    assert st["exact_queries"] == 0 and st["pool_overflows"] == 0, st
    assert torch.all(counts == k)
    assert torch.all(dists[:, 1:] >= dists[:, :-1])                                                # ascending (-dot)
    assert torch.all((ids >= first_id) & (ids < first_id + n))                                     # global ids of THIS shard
    srt = torch.sort(ids, dim=1).values
    assert torch.all(srt[:, 1:] != srt[:, :-1])                                                    # no duplicate ids
    ids2, dists2, _ = search(queries, k)                                                           # idempotence
    assert torch.equal(ids, ids.clone()) and torch.equal(ids2, ids) and torch.equal(dists2, dists)
    # the k = 10 answer is a prefix of the k = 100 answer
    i10, d10, _ = (t.clone() for t in search(queries[:256].contiguous(), 10))
    i100, d100, _ = search(queries[:256].contiguous(), k)
    assert torch.equal(i10, i100[:, :10]) and torch.equal(d10, d100[:, :10])
    rows_h, q_h = rows.cpu().numpy(), queries.cpu().numpy()
    ids_h, dists_h = ids2.cpu().numpy().astype(np.uint64), dists2.cpu().numpy()
    gid = np.arange(first_id, first_id + n, dtype=np.uint64)
    for b in (0, 777):                                                                             # one query of pass 0 (block 0), one of pass 1 (block 1)
        oi, od = oracle.flat_search(2, rows_h, q_h[b], k, ids=gid)
        assert np.array_equal(ids_h[b], oi) and np.array_equal(dists_h[b].view(np.uint32), od.view(np.uint32)), b


def test_c4_1m_x_1536_euclid_with_the_compiled_string_filter(vdb):
    n, d, B, k = 1_000_000, 1536, 256, 10
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    ix.reserve(n, d)
    rows_h = np.empty((n, d), dtype=np.float32)
    for c in range(4):                                                                             # 4 x 1.5 GB blocks
        blk = torch.rand((n // 4, d), generator=g, device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        ix.add_bulk_device(blk.data_ptr(), n // 4, d, first_id=c * (n // 4))
        rows_h[c * (n // 4):(c + 1) * (n // 4)] = blk.cpu().numpy()
        del blk
    ix.flush()
    g.manual_seed(6)
    queries = torch.rand((B, d), generator=g, device=dev, dtype=torch.float32)
    q_h = queries.cpu().numpy()
    # metadata: color = palette[row mod 4] as STRINGS, compiled to the device bitmask
    palette = np.array(["red", "green", "blue", "amber"], dtype=object)
    store = vdb.VectorStore.with_index(ix)
    store.attach_bulk_metadata(n, {"color": palette[np.arange(n) % 4]})
    mask, bits = store.compile_filter(vdb.MetadataFilter.Eq("color", "red"))
    assert bits == n and int(np.unpackbits(mask.view(np.uint8)).sum()) == n // 4
    gi, gd, gc = ix.search_batch_arrays(q_h, k, id_mask=mask, mask_bits=bits)
    st = ix.last_stats()
    assert st["bf16_screen"] == 1 and st["exact_queries"] == 0 and st["pool_overflows"] == 0, st
    assert np.all(gc == k) and np.all(gi % 4 == 0)                                                 # only "red" rows
    assert np.all(gd[:, 1:] >= gd[:, :-1])
    live = (np.arange(n) % 4 == 0).astype(np.uint8)
    for b in (0, 200):
        oi, od = oracle.flat_search(0, rows_h, q_h[b], k, live=live)
        assert np.array_equal(gi[b], oi) and np.array_equal(gd[b].view(np.uint32), od.view(np.uint32)), b
    # the reference post-filters a 3k over-fetch (storage.rs:268-287): its result is a PREFIX of the pre-filtered one
    fi, fd, fc = ix.search_batch_arrays(q_h[:32], 3 * k)
    for b in range(32):
        post = [int(i) for i in fi[b, :fc[b]] if i % 4 == 0][:k]
        assert [int(i) for i in gi[b][:len(post)]] == post, b
    # the same through the store's own batch entry point (string ids out)
    res = store.search_batch_prefiltered([(vdb.Vector(q_h[b]), k) for b in (0, 1)], vdb.MetadataFilter.Eq("color", "red"))
    assert [r.id for r in res[0]] == [str(int(i)) for i in gi[0]]
    # an And / Ne combination: not red and not blue = green or amber
    F = vdb.MetadataFilter
    mask2, _ = store.compile_filter(F.And([F.Ne("color", "red"), F.Ne("color", "blue")]))
    hi, _, hc = ix.search_batch_arrays(q_h[:8], k, id_mask=mask2, mask_bits=bits)
    assert np.all(hc == k) and np.all((hi % 4 == 1) | (hi % 4 == 3))


def test_c3_full_size_10m_rows_as_eight_shards_of_one_handle(vdb):
    """BASELINE configs[2] at FULL size: 10M x 768 f32 (30.7 GB of the GPU's 288 GB), dot, batch 1024, k = 100, as ONE
    vdb_flat_create_sharded handle with 8 row shards on the one GPU of the box -- everything of the 8-GPU job but the wires:
    rows dealt to the shards, eight local searches of two 512-query passes each, one packed exchange, the merge.  Two queries
    are compared with the oracle over all 10M rows (ids, order, distance bits)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("c3_full_rehearsal", os.path.join(root, "tools", "c3_full_rehearsal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if torch.cuda.mem_get_info()[0] < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free device memory")
    mod.main()


def test_c5_hnsw_at_768_dimensions_against_the_cpu_restatement(vdb):
    """BASELINE config 5 at a build-affordable size: 30k x 768, m = 16, ef_construction = 100, ef_search = 200, batch 256.
    Same seed and insertion order -> the GPU-offloaded index must hold the CPU restatement's graph and return its results."""
    n, d, m, efc, efs, B, k = 30_000, 768, 16, 100, 200, 256, 10
    rng = np.random.default_rng(55)
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((B, d), dtype=np.float32)
    gidx = vdb.GpuHnswIndex(vdb.DistanceMetric.Euclidean, vdb.HnswParams.new(m, efc, 50), seed=11)
    gidx.build_batch((np.arange(n, dtype=np.uint64), rows))
    o = oracle.HnswOracle(0, m=m, ef_construction=efc, ef_search=50, seed=11)
    for i in range(n):
        o.insert(i, rows[i])
    assert gidx.len() == len(o) == n and gidx.entry_point() == o.entry_point()
    for i in list(range(0, n, 97)) + [n - 1]:                                                      # every 97th node: levels and all lists
        lv = o.level(i)
        assert gidx.level(i) == lv
        for l in range(lv + 1):
            assert gidx.neighbors(i, l) == o.neighbors(i, l), (i, l)
    gi, gd, gc = gidx.search_batch_arrays(q, k, efs)
    st = gidx.stats()
    assert st["device_queries"] == B and st["host_redone"] == 0, st                                # device-resident walks
    for b in range(0, B, 8):
        oi, od = o.search(q[b], k, efs)
        assert gc[b] == len(oi) and np.array_equal(gi[b, :gc[b]], oi) and np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)), b
    # recall@10 of the graph search against the exact search (the reference's recall test uses FlatIndex as ground truth,
    # tests/recall_test.rs:18-26); the number is recorded, the floor only guards against a broken graph
    flat = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    flat.add_bulk(rows)
    fi, _, _ = flat.search_batch_arrays(q, k)
    rec = float(np.mean([oracle.recall(fi[b], gi[b, :gc[b]]) for b in range(B)]))
    print(f"\n[c5] HNSW 30k x 768 m=16 ef=200: recall@10 vs exact = {rec:.3f}")
    assert rec > 0.3


def test_c5_hnsw_200k_x_768_frontier_only_build(vdb):
    """BASELINE config 5's parameters (m = 16, ef_construction = 200, ef_search = 200, 768 dimensions) at 200,000 rows -- a size
    the CPU restatement cannot follow inside a test suite (hours), so the checks are size-independent: the build evaluates about
    what the reference's algorithm consumes (not N^2 / 2 = 2e10), every list respects the reference's caps (graph.rs:40,
    :207-241), the levels follow the seeded stream, the device-resident search equals the host traversal of the same graph bit
    for bit."""
    import time
    n, d, m, efc, efs, B, k = 200_000, 768, 16, 200, 200, 64, 10
    rng = np.random.default_rng(77)
    rows = rng.random((n, d), dtype=np.float32)
    g = vdb.GpuHnswIndex(vdb.DistanceMetric.Euclidean, vdb.HnswParams.new(m, efc, 50), seed=5)
    t0 = time.perf_counter()
    for c0 in range(0, n, 50_000):
        g.build_batch((np.arange(c0, c0 + 50_000, dtype=np.uint64), rows[c0:c0 + 50_000]))
    t_build = time.perf_counter() - t0
    bs = g.build_stats()
    gpu = bs["walk_distances"] + bs["in_chunk_distances"] + bs["miss_distances"]
    print(f"\n[c5] HNSW 200k x 768 build {t_build:.1f} s; GPU distances {gpu:.3e} = {gpu / bs['reference_distances']:.3f} x the reference "
          f"algorithm's {bs['reference_distances']:.3e} (row scans: {n * (n - 1) / 2:.1e}); misses {bs['miss_round_trips']}")
    assert bs["frontier_inserts"] == n and bs["scan_inserts"] == 0, bs
    assert gpu <= 1.5 * bs["reference_distances"] and gpu < 0.1 * n * (n - 1) / 2, bs
    assert g.len() == n
    ep, max_level = g.entry_point()
    assert g.level(ep) == max_level
    for i in list(range(0, n, 4001)) + [ep]:
        lv = g.level(i)
        for l in range(lv + 1):
            nb = g.neighbors(i, l)
            assert len(nb) <= (2 * m if l == 0 else m) and len(set(nb)) == len(nb) and i not in nb, (i, l)
            assert all(g.level(int(x)) >= l for x in nb), (i, l)
    q = rng.random((B, d), dtype=np.float32)
    di, dd, dc = g.search_batch_arrays(q, k, efs)
    assert g.stats()["device_queries"] == B and g.stats()["host_redone"] == 0
    g.set_traversal(host_only=True)
    hi, hd, hc = g.search_batch_arrays(q[:16], k, efs)
    g.set_traversal(host_only=False)
    assert np.array_equal(hc, dc[:16]) and np.array_equal(hi, di[:16]) and np.array_equal(hd.view(np.uint32), dd[:16].view(np.uint32))
    assert np.all(dd[:, 1:] >= dd[:, :-1]) and np.all(dc == k)
    # (no self-query assertion: on uniform 768-dimensional data the reference's simple neighbour selection reaches recall@10 of
    # ~0.3 at this size -- DESIGN.md 11 -- and a walk from the entry point need not arrive at a stored row that is the query)
