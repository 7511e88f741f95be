"""The GPU-offloaded HNSW index (include/vdb_hnsw.h, BASELINE config 5) against the CPU restatement of the
reference's HNSW (oracle/hnsw_oracle.c).  With the same seed and insertion order the two must hold the SAME graph
(entry point, node levels, every neighbour list in order) and return the SAME results (ids, order, distance bits):
the product runs the reference's traversal and only moves the distance evaluations to the GPU.  Against the
reference itself parity is statistical (its levels are unseeded): the recall floors of tests/recall_test.rs."""
import numpy as np
import pytest

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vdb():
    v = load_package()
    v.build()
    return v


def build_pair(vdb, metric, rows, m, efc, efs, seed, bulk=True, ids=None):
    g = vdb.GpuHnswIndex(vdb.DistanceMetric(metric), vdb.HnswParams.new(m, efc, efs), seed=seed)
    o = oracle.HnswOracle(metric, m=m, ef_construction=efc, ef_search=efs, seed=seed)
    ids = np.arange(rows.shape[0], dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
    if bulk:
        g.build_batch((ids, rows))
    else:
        for i, v in zip(ids, rows):
            g.add(int(i), vdb.Vector(v))
    for i, v in zip(ids, rows):
        o.insert(int(i), v)
    return g, o, ids


def assert_same_graph(g, o, ids):
    assert g.len() == len(o)
    assert g.entry_point() == o.entry_point()
    for i in ids:
        lv = o.level(int(i))
        assert g.level(int(i)) == lv, i
        if lv < 0:
            continue
        for l in range(lv + 1):
            assert g.neighbors(int(i), l) == o.neighbors(int(i), l), (int(i), l)
        assert g.neighbors(int(i), lv + 1) is None


def assert_same_results(g, o, queries, k, ef):
    gi, gd, gc = g.search_batch_arrays(queries, k, ef)
    for b in range(queries.shape[0]):
        oi, od = o.search(queries[b], k, ef)
        assert gc[b] == len(oi), (b, gc[b], len(oi))
        assert np.array_equal(gi[b, :gc[b]], oi), (b, gi[b, :gc[b]], oi)
        assert np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)), (b, gd[b, :gc[b]], od)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,m,efc", [(300, 8, 4, 32), (1500, 48, 16, 200), (800, 100, 8, 64)])
def test_graph_and_results_equal_the_cpu_restatement(vdb, metric, n, d, m, efc):
    rng = np.random.default_rng(n + d + metric)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((37, d)).astype(np.float32)
    g, o, ids = build_pair(vdb, metric, rows, m, efc, 50, seed=7 + metric)
    assert_same_graph(g, o, ids)
    assert_same_results(g, o, queries, 10, 100)
    assert_same_results(g, o, queries[:5], 1, 16)
    assert_same_results(g, o, queries[:5], 200, 50)                     # k > ef: ef_actual = k (graph.rs:406)
    # the result heap at the edge of what a wave pops with four ballots (514 entries) and beyond it (single-lane pop between wave pushes)
    assert_same_results(g, o, queries[:4], 10, 513)
    assert_same_results(g, o, queries[:4], 10, 700)
    st = g.stats()
    assert st["gpu_distances"] > 0 and st["device_queries"] == 37 + 5 + 5 + 4 + 4 and st["host_redone"] == 0, st


def test_device_resident_and_host_traversal_agree(vdb):
    """The same batch through the device-resident search (default) and the host traversal: both are the reference's walk."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(12)
    rows = rng.random((3000, 40), dtype=np.float32)
    q = rng.random((64, 40), dtype=np.float32)
    g, o, ids = build_pair(vdb, 0, rows, 16, 100, 50, seed=2)
    gi, gd, gc = g.search_batch_arrays(q, 10, 200)
    assert g.stats()["device_queries"] == 64
    # m = 24 (lists of up to 49 neighbours) does not fit the device kernel: host traversal, same answers as the oracle
    g2, o2, _ = build_pair(vdb, 0, rows, 24, 100, 50, seed=2)
    assert_same_results(g2, o2, q[:9], 10, 64)
    assert g2.stats()["device_queries"] == 0 and g2.stats()["last_search_rounds"] > 0
    # the host traversal forced for the same batch (vdb_hnsw_set_traversal): identical results
    g.set_traversal(host_only=True, host_threads=3)
    hi, hd, hc = g.search_batch_arrays(q, 10, 200)
    assert g.stats()["device_queries"] == 64 and g.stats()["last_search_rounds"] > 0
    assert np.array_equal(hi, gi) and np.array_equal(hd.view(np.uint32), gd.view(np.uint32)) and np.array_equal(hc, gc)
    g.set_traversal(host_only=False)


def test_single_adds_equal_bulk_build(vdb):
    rng = np.random.default_rng(5)
    rows = rng.random((400, 12), dtype=np.float32)
    a, o, ids = build_pair(vdb, 0, rows, 8, 64, 50, seed=3, bulk=False)
    b, _, _ = build_pair(vdb, 0, rows, 8, 64, 50, seed=3, bulk=True)
    assert_same_graph(a, o, ids)
    assert_same_graph(b, o, ids)


def test_duplicate_rows_tie_order_follows_the_heap_layout(vdb):
    """Equal distances: into_sorted_vec keeps the BinaryHeap's array order among ties (neighbor_queue.rs:102-106);
    the product and the restatement must agree on it."""
    rng = np.random.default_rng(6)
    base = rng.random((60, 6), dtype=np.float32)
    rows = np.concatenate([base] * 5, 0)                                   # every vector 5 times
    g, o, ids = build_pair(vdb, 0, rows, 6, 40, 50, seed=9)
    assert_same_graph(g, o, ids)
    assert_same_results(g, o, base[:20], 12, 64)


def test_remove_semantics(vdb):
    rng = np.random.default_rng(8)
    rows = rng.random((500, 10), dtype=np.float32)
    g, o, ids = build_pair(vdb, 0, rows, 8, 64, 50, seed=4)
    ep, _ = o.entry_point()
    for victim in [3, 77, 78, ep, 499, 12345]:                             # incl. the entry point and an absent id
        g.remove(victim)
        o.remove(victim)
    assert_same_graph(g, o, ids)
    q = rng.random((9, 10), dtype=np.float32)
    assert_same_results(g, o, q, 10, 50)
    # inserts after deletions keep following the same random stream
    extra = rng.random((40, 10), dtype=np.float32)
    for j, v in enumerate(extra):
        g.add(1000 + j, vdb.Vector(v))
        o.insert(1000 + j, v)
    assert_same_graph(g, o, list(ids) + list(range(1000, 1040)))
    assert_same_results(g, o, q, 10, 50)


def test_reference_unit_tests_through_the_trait(vdb):
    V, M = vdb.Vector, vdb.DistanceMetric
    ix = vdb.GpuHnswIndex(M.Euclidean)                                     # mod.rs:88-98
    ix.add(0, V([1.0, 0.0, 0.0])); ix.add(1, V([0.0, 1.0, 0.0])); ix.add(2, V([1.0, 1.0, 0.0]))
    res = ix.search(V([1.0, 0.0, 0.0]), 2)
    assert len(res) == 2 and res[0][0] == 0 and res[0][1] < 1e-5
    assert ix.get_vector(0) == V([1.0, 0.0, 0.0]) and ix.get_vector(99) is None       # mod.rs:101-108
    g = vdb.GpuHnswIndex(M.Euclidean, vdb.HnswParams.new(4, 32, 16))       # graph.rs:488-504
    for i in range(5):
        g.add(i, V([float(i), 0.0]))
    r = g.search_with_ef(V([0.5, 0.0]), 2, 16)
    assert len(r) == 2 and {r[0][0], r[1][0]} == {0, 1}
    g = vdb.GpuHnswIndex(M.Euclidean, vdb.HnswParams.new(4, 32, 16))       # graph.rs:507-537
    g.add(0, V([1.0, 0.0])); g.add(1, V([0.0, 1.0])); g.add(2, V([1.0, 1.0]))
    ep, _ = g.entry_point()
    g.remove(ep)
    assert g.len() == 2 and len(g.search_with_ef(V([0.0, 1.0]), 1, 16)) == 1
    assert vdb.GpuHnswIndex(M.Cosine).search(V([1.0, 2.0]), 3) == []       # empty graph (graph.rs:392-395)


def test_hnsw_behind_the_vector_store(vdb):
    # mod.rs:111-133 / :136-153: VectorStore::with_index(HnswIndex)
    V = vdb.Vector
    store = vdb.VectorStore.with_index(vdb.GpuHnswIndex(vdb.DistanceMetric.Euclidean, vdb.HnswParams.new(4, 32, 16)))
    store.insert("v1", V([1.0, 0.0, 0.0])); store.insert("v2", V([0.0, 1.0, 0.0])); store.insert("v3", V([0.0, 0.0, 1.0]))
    res = store.search(V([1.0, 0.1, 0.0]), 2)
    assert len(res) == 2 and res[0].id == "v1"
    store.delete("v1")
    assert len(store) == 2


def test_error_semantics(vdb):
    V = vdb.Vector
    ix = vdb.GpuHnswIndex(vdb.DistanceMetric.Cosine, vdb.HnswParams.new(4, 32, 16))
    ix.add(0, V([1.0, 0.0])); ix.add(1, V([0.0, 1.0]))
    with pytest.raises(vdb.InvalidVector):                                 # zero-norm query (distance.rs:51-55)
        ix.search(V([0.0, 0.0]), 1)
    with pytest.raises(vdb.DimensionMismatch):
        ix.search(V([1.0, 0.0, 0.0]), 1)
    with pytest.raises(vdb.DimensionMismatch):
        ix.add(2, V([1.0, 0.0, 0.0]))


@pytest.mark.parametrize("n,dim,nq,floor", [(100, 32, 50, 0.90), (1000, 64, 50, 0.90), (5000, 128, 20, 0.85)])
def test_recall_floors_of_the_reference(vdb, n, dim, nq, floor):
    # tests/recall_test.rs:28-80 with the GPU FlatIndex as ground truth
    rng = np.random.default_rng(n)
    rows = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    flat = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    flat.add_bulk(rows)
    hnsw = vdb.GpuHnswIndex(vdb.DistanceMetric.Euclidean, vdb.HnswParams.new(16, 200, 50), seed=n)
    hnsw.build_batch((np.arange(n, dtype=np.uint64), rows))
    ti, _, _ = flat.search_batch_arrays(queries, 10)
    hi, hd, hc = hnsw.search_batch_arrays(queries, 10, 100)
    rec = np.mean([len(set(ti[b]) & set(hi[b, :hc[b]])) / 10.0 for b in range(nq)])
    assert rec >= floor, rec
    assert np.all(hd[:, 1:] >= hd[:, :-1])


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_frontier_only_build_equals_the_scan_build_and_the_cpu_restatement(vdb, metric):
    """bulk inserts used to fold every new vector against every stored row (N^2 / 2 distances).  The default
    build now runs a device walk per insert that evaluates only what search_layer asks for (graph.rs:155, :182) on the graph as
    of a block of 64 inserts earlier (the walks run one block ahead of the host), and the host replays the inserts in order -- the SAME graph, node for node, as the
    row-scan build and as the CPU restatement; the GPU evaluates a small multiple of the distances the reference's algorithm
    consumes, not N^2 / 2."""
    rng = np.random.default_rng(90 + metric)
    n, d, m, efc = 6000, 64, 12, 100
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[3000:3040] = rows[100:140]                                    # duplicates: exact ties inside the walks
    g, o, ids = build_pair(vdb, metric, rows, m, efc, 50, seed=11 + metric)          # default: frontier only
    assert_same_graph(g, o, ids)
    bs = g.build_stats()
    assert bs["frontier_inserts"] == n and bs["scan_inserts"] == 0, bs
    gpu = bs["walk_distances"] + bs["in_chunk_distances"] + bs["miss_distances"]
    assert bs["reference_distances"] > 0 and gpu <= 3 * bs["reference_distances"], bs     # within 3x of what the algorithm evaluates
    assert gpu < n * (n - 1) // 2, bs                                                   # (the row scans' count; at 6000 nodes a walk still sees a sixth of the graph)
    assert bs["record_overflows"] == 0, bs
    bt = g.build_times()                                               # vdb_hnsw_build_times: where the build's wall time went
    assert bt["replay_s"] > 0 and bt["replay_s"] >= bt["replay_miss_round_trips_s"] >= 0 and bt["walks_s"] >= 0 and bt["mirror_sync_s"] >= 0, bt
    g2 = vdb.GpuHnswIndex(vdb.DistanceMetric(metric), vdb.HnswParams.new(m, efc, 50), seed=11 + metric)
    g2.set_build(False)                                                # the row-scan build of round 2
    g2.build_batch((ids, rows))
    assert g2.build_stats()["scan_inserts"] == n
    assert g2.entry_point() == g.entry_point()
    for i in range(0, n, 7):
        lv = g.level(i)
        assert g2.level(i) == lv
        for l in range(lv + 1):
            assert g2.neighbors(i, l) == g.neighbors(i, l), (i, l)
    queries = rng.standard_normal((20, d)).astype(np.float32)
    assert_same_results(g, o, queries, 10, 100)
    # a second bulk on top of the first (the mirror grows incrementally), single adds in between, a removal (full rebuild)
    more = rng.standard_normal((700, d)).astype(np.float32)
    g.build_batch((np.arange(n, n + 700, dtype=np.uint64), more))
    for i, v in enumerate(more):
        o.insert(n + i, v)
    g.add(n + 700, vdb.Vector(rows[5])); o.insert(n + 700, rows[5])
    g.remove(17); o.remove(17)
    extra = rng.standard_normal((64, d)).astype(np.float32)
    g.build_batch((np.arange(n + 701, n + 765, dtype=np.uint64), extra))
    for i, v in enumerate(extra):
        o.insert(n + 701 + i, v)
    all_ids = np.arange(n + 765, dtype=np.uint64)
    assert_same_graph(g, o, all_ids)
    assert_same_results(g, o, queries, 10, 100)


def test_frontier_only_build_reports_a_zero_norm_row_like_the_reference(vdb):
    """mod.rs:37-42: build_batch returns at the first failing insert; that node is stored without links, later vectors are not."""
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((200, 16)).astype(np.float32)
    rows[120] = 0.0
    g = vdb.GpuHnswIndex(vdb.DistanceMetric.Cosine, vdb.HnswParams.new(8, 40, 20), seed=3)
    o = oracle.HnswOracle(1, m=8, ef_construction=40, ef_search=20, seed=3)
    with pytest.raises(vdb.InvalidVector):
        g.build_batch((np.arange(200, dtype=np.uint64), rows))
    failed = False
    for i, v in enumerate(rows):
        try:
            o.insert(i, v)
        except Exception:
            failed = True
            break
    assert failed and g.len() == len(o) == 121
    assert_same_graph(g, o, np.arange(120, dtype=np.uint64))
    # the level generator is where the reference's would be: the next inserts draw the same levels
    g.remove(120); o.remove(120)
    nxt = rng.standard_normal((40, 16)).astype(np.float32)
    g.build_batch((np.arange(300, 340, dtype=np.uint64), nxt))
    for i, v in enumerate(nxt):
        o.insert(300 + i, v)
    assert_same_graph(g, o, np.concatenate([np.arange(120), np.arange(300, 340)]).astype(np.uint64))
