"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Integer results (ids, counts,
order) must match exactly; distances are recomputed on the GPU in the reference's operation order
and must be BIT-IDENTICAL to the oracle's (tolerance 0)."""
import os

import numpy as np
import pytest

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu

M = {"euclidean": 0, "cosine": 1, "dot": 2}


@pytest.fixture(scope="module")
def vdb():
    v = load_package()
    v.build()
    return v


def make_index(vdb, metric, rows, ids=None):
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    ix.add_bulk(rows, ids=ids)
    return ix


def check_against_oracle(vdb, metric, rows, queries, k, ids=None, live=None, ix=None, qsel=None):
    ix = ix or make_index(vdb, metric, rows, ids)
    gi, gd, gc = ix.search_batch_arrays(queries, k)
    qsel = range(queries.shape[0]) if qsel is None else qsel
    for b in qsel:
        oi, od = oracle.flat_search(metric, rows, queries[b], k, ids=ids, live=live)
        assert gc[b] == len(oi), (b, gc[b], len(oi))
        assert np.array_equal(gi[b, :gc[b]], oi), (b, gi[b, :gc[b]], oi, gd[b, :gc[b]], od)
        assert np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)), (b, gd[b, :gc[b]], od)
    return ix


# ------------------------------------------------------------------ reference known answers
def test_known_flat_search(vdb, known_answers):
    for c in known_answers["flat_search"]:
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(M[c["metric"]]))
        for i, v in c["rows"].items():
            ix.add(int(i), vdb.Vector(v))
        res = ix.search(vdb.Vector(c["query"]), c["k"])
        assert len(res) == c["expect_len"], c["src"]
        assert res[0][0] == c["expect_first_id"], c["src"]
        if "expect_first_dist_lt" in c:
            assert res[0][1] < c["expect_first_dist_lt"]


def test_known_distances_via_single_row_index(vdb, known_answers):
    for c in known_answers["distance"]:
        metric = "dot" if c["metric"] == "dot_raw" else c["metric"]
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(M[metric]))
        ix.add(0, vdb.Vector(c["b"]))
        (rid, d), = ix.search(vdb.Vector(c["a"]), 1)
        d = -d if c["metric"] == "dot_raw" else d
        assert abs(float(d) - c["expect"]) <= c["eps"] * max(1.0, abs(c["expect"])), c["src"]
        assert float(d) == float((-1 if c["metric"] == "dot_raw" else 1) * oracle.distance(M[metric], c["a"], c["b"]))


def test_known_store_cases(vdb, known_answers):
    V, S = vdb.Vector, vdb.VectorStore
    for c in known_answers["empty_store"]:
        assert S(M[c["metric"]]).search(V(c["query"]), c["k"]) == []
    for c in known_answers["remove"]:
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(M[c["metric"]]))
        for i, v in c["rows"].items():
            ix.add(int(i), V(v))
        assert ix.len() == c["expect_len_before"]
        for i in c["remove"]:
            ix.remove(i)
        assert ix.len() == c["expect_len_after"]
    for c in known_answers["filter"]:
        st = S(M[c["metric"]])
        for name, v in c["rows"].items():
            st.insert_with_metadata(name, V(v), vdb.Metadata(c["meta"][name]))
        f = vdb.MetadataFilter.Eq(c["filter"]["field"], c["filter"]["value"])
        got = st.search_with_filter(V(c["query"]), c["k"], f)
        assert sorted(r.id for r in got) == sorted(c["expect_id_set"]), c["src"]
        pre = st.search_batch_prefiltered([(V(c["query"]), c["k"])], f)[0]
        assert [r.id for r in pre][:len(got)] == [r.id for r in got]      # reference result is a prefix
    for c in known_answers["batch"]:
        st = S(M[c["metric"]])
        for name, v in c["rows"].items():
            st.insert(name, V(v))
        res = st.search_batch([(V(q), k) for q, k in c["queries"]])
        assert [[r.id for r in rs] for rs in res] == c["expect_ids"]
    for c in known_answers["batch_filter"]:
        st = S(M[c["metric"]])
        for name, v in c["rows"].items():
            st.insert_with_metadata(name, V(v), vdb.Metadata(c["meta"][name]))
        f = vdb.MetadataFilter.Eq(c["filter"]["field"], c["filter"]["value"])
        res = st.search_batch_with_filter([(V(q), k) for q, k in c["queries"]], f)
        assert [[r.id for r in rs] for rs in res] == c["expect_ids"]
    for c in known_answers["insert_errors"]:
        st = S(0)
        st.insert("v1", V(c["first"]))
        with pytest.raises(vdb.DimensionMismatch):
            st.insert("v2", V(c["second"]))


# ------------------------------------------------------------------ committed golden vectors
def test_golden_vectors(vdb, golden_cases):
    g = golden_cases
    for name in sorted({k.split("/")[0] for k in g.files}):
        rows, queries, ids = g[f"{name}/rows"], g[f"{name}/queries"], g[f"{name}/ids"]
        for mname, m in M.items():
            ix = None
            for k in g[f"{name}/ks"]:
                key = f"{name}/{mname}/k{k}/ids"
                if key not in g.files:
                    continue
                ix = ix or make_index(vdb, m, rows, ids)
                gi, gd, gc = ix.search_batch_arrays(queries, int(k))
                eid, ed = g[key], g[f"{name}/{mname}/k{k}/dists"]
                assert np.all(gc == eid.shape[1]), (name, mname, k)
                assert np.array_equal(gi[:, :eid.shape[1]], eid), (name, mname, k)
                assert np.array_equal(gd[:, :eid.shape[1]].view(np.uint32), ed.view(np.uint32)), (name, mname, k)


# ------------------------------------------------------------------ seeded sweeps vs the oracle
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq,k", [(1, 3, 1, 1), (31, 5, 3, 4), (1000, 33, 17, 10), (5000, 128, 40, 10)])
def test_small_index_paths(vdb, metric, n, d, nq, k):
    rng = np.random.default_rng(n * 7 + d)
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((nq, d), dtype=np.float32)
    check_against_oracle(vdb, metric, rows, q, k)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq,k", [(20000, 64, 33, 10), (50000, 128, 70, 10), (40000, 50, 130, 25),
                                      (70001, 96, 256, 10), (30000, 32, 5, 100)])
def test_fused_path_vs_oracle(vdb, metric, n, d, nq, k):
    rng = np.random.default_rng(n + d + nq)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = check_against_oracle(vdb, metric, rows, q, k, qsel=range(0, nq, max(1, nq // 12)))
    st = ix.last_stats()
    assert st["rows_scanned"] >= n and st["sample_rows"] > 0, st      # the fused MFMA kernel ran
    assert st["pool_overflows"] == 0, st


def test_uniform_concentrated_data(vdb):
    # the reference benches' distribution: uniform[0,1) (benches/search_bench.rs:6-13)
    rng = np.random.default_rng(5)
    rows = rng.random((120000, 128), dtype=np.float32)
    q = rng.random((64, 128), dtype=np.float32)
    q[0] = 0.5                                                         # the bench query [0.5; 128]
    for metric in (0, 1, 2):
        ix = check_against_oracle(vdb, metric, rows, q, 10, qsel=range(0, 64, 8))
        st = ix.last_stats()
        assert st["exact_queries"] == 0, st                            # every query certified on the MFMA path


def test_forced_exact_fallback_matches(vdb):
    rng = np.random.default_rng(11)
    rows = rng.standard_normal((30000, 40)).astype(np.float32)
    q = rng.standard_normal((9, 40)).astype(np.float32)
    ix = make_index(vdb, 0, rows)
    a = ix.search_batch_arrays(q, 10)
    ix.set_tiers(ix.TIERS_FORCE_EXACT)
    b = ix.search_batch_arrays(q, 10)
    assert ix.last_stats()["exact_queries"] == 9
    ix.set_tiers(0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    check_against_oracle(vdb, 0, rows, q, 10, ix=ix)


def test_large_k_uses_exact_path(vdb):
    rng = np.random.default_rng(12)
    rows = rng.random((3000, 24), dtype=np.float32)
    q = rng.random((4, 24), dtype=np.float32)
    for metric in (0, 1, 2):
        check_against_oracle(vdb, metric, rows, q, 500)
        check_against_oracle(vdb, metric, rows, q, 5000)               # k > n returns n results


def test_duplicates_and_near_ties_are_certified_or_fall_back(vdb):
    rng = np.random.default_rng(13)
    base = rng.random((500, 48), dtype=np.float32)
    rows = np.concatenate([base] * 80, 0)                              # 40000 rows, every row 80 times
    ids = rng.permutation(rows.shape[0]).astype(np.uint64)             # ids not monotone in row order
    q = base[:6] + 0.0
    for metric in (0, 1, 2):
        check_against_oracle(vdb, metric, rows, q, 10, ids=ids)


# ------------------------------------------------------------------ mutation semantics
def test_remove_overwrite_and_tombstones(vdb):
    rng = np.random.default_rng(14)
    n, d = 26000, 20
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((7, d), dtype=np.float32)
    ix = make_index(vdb, 0, rows)
    live = np.ones(n, dtype=np.uint8)
    for r in rng.choice(n, 3000, replace=False):
        ix.remove(int(r))
        live[r] = 0
    ix.remove(10 ** 9)                                                 # absent id is Ok (flat_index.rs:43-46)
    assert ix.len() == int(live.sum())
    check_against_oracle(vdb, 0, rows, q, 10, live=live, ix=ix)
    # overwrite an id with new data (HashMap::insert, flat_index.rs:39)
    rows2 = rows.copy()
    for r in (5, 77, 4000):
        rows2[r] = q[0] + 1e-3 * (r % 7)
        ix.add(r, vdb.Vector(rows2[r]))
        live[r] = 1
    assert ix.len() == int(live.sum())
    check_against_oracle(vdb, 0, rows2, q, 10, live=live, ix=ix)


def test_prefilter_mask_matches_oracle_and_reference_prefix(vdb):
    rng = np.random.default_rng(15)
    n, d, k = 40000, 32, 10
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((5, d), dtype=np.float32)
    ix = make_index(vdb, 0, rows)
    keep = (np.arange(n) % 4 == 0)                                     # 25% selectivity
    mask = np.zeros((n + 63) // 64, dtype=np.uint64)
    for i in np.nonzero(keep)[0]:
        mask[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    gi, gd, gc = ix.search_batch_arrays(q, k, id_mask=mask, mask_bits=n)
    for b in range(5):
        oi, od = oracle.flat_search(0, rows, q[b], k, live=keep.astype(np.uint8))
        assert np.array_equal(gi[b, :gc[b]], oi) and np.array_equal(gd[b, :gc[b]], od)
        ri, rd = oracle.search_with_filter(0, rows, q[b], k, keep.astype(np.uint8))   # reference post-filter
        assert np.array_equal(gi[b, :len(ri)], ri)                     # ... is a prefix of the pre-filter


# ------------------------------------------------------------------ error semantics
def test_error_semantics(vdb):
    V = vdb.Vector
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Cosine)
    ix.add(0, V([1, 0, 0]))
    ix.add(1, V([0, 0, 0]))                                            # zero row is accepted at add ...
    with pytest.raises(vdb.InvalidVector):                             # ... and fails every search
        ix.search(V([1, 1, 0]), 1)
    ix.remove(1)
    assert ix.search(V([1, 1, 0]), 1)[0][0] == 0
    with pytest.raises(vdb.InvalidVector):
        ix.search(V([0, 0, 0]), 1)                                     # zero query
    with pytest.raises(vdb.DimensionMismatch) as e:
        ix.search(V([1, 0]), 1)
    assert (e.value.expected, e.value.actual) == (2, 3)                # distance.rs:22-25
    ix.add(5, V([1, 2]))                                               # no dimension check at add (flat_index.rs:38)
    with pytest.raises(vdb.DimensionMismatch):
        ix.search(V([1, 0, 0]), 1)
    ix.remove(5)
    assert len(ix.search(V([1, 0, 0]), 3)) == 1
    e2 = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean)
    assert e2.search(V([1, 2, 3]), 5) == []                            # empty index
    e2.add(0, V([float("nan"), 1.0]))
    e2.add(1, V([0.0, 1.0]))
    with pytest.raises(vdb.NanDistance):                               # reference panics (flat_index.rs:62)
        e2.search(V([0.0, 0.0]), 1)
    assert e2.search(V([1.0, 1.0]), 0) == []


def test_per_query_k_and_ragged_batch(vdb):
    rng = np.random.default_rng(16)
    rows = rng.random((900, 12), dtype=np.float32)
    ix = make_index(vdb, 2, rows)
    q = rng.random((6, 12), dtype=np.float32)
    ks = np.array([1, 10, 0, 3, 950, 7])
    gi, gd, gc = ix.search_batch_arrays(q, ks)
    for b in range(6):
        oi, od = oracle.flat_search(2, rows, q[b], int(ks[b]))
        assert gc[b] == len(oi) and np.array_equal(gi[b, :gc[b]], oi) and np.array_equal(gd[b, :gc[b]], od)


def test_recall_is_one(vdb):
    rng = np.random.default_rng(17)
    rows = rng.random((60000, 64), dtype=np.float32)
    q = rng.random((32, 64), dtype=np.float32)
    ix = make_index(vdb, 1, rows)
    gi, _, gc = ix.search_batch_arrays(q, 10)
    rec = [oracle.recall(oracle.flat_search(1, rows, q[b], 10)[0], gi[b, :gc[b]]) for b in range(0, 32, 4)]
    assert min(rec) == 1.0


def test_mmap_vector_file_bulk_load(vdb, tmp_path):
    """The reference's mmap vector file (src/persistence/mmap.rs:13-15,77-84): 8-byte header
    [dim u32 LE][count u32 LE] + row-major LE f32 rows, loaded in one call."""
    rng = np.random.default_rng(21)
    n, d = 20001, 37
    rows = rng.random((n, d), dtype=np.float32)
    path = tmp_path / "vectors.bin"
    with open(path, "wb") as f:
        f.write(np.array([d, n], dtype="<u4").tobytes())
        f.write(rows.astype("<f4").tobytes())
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    assert ix.load_vector_file(str(path), first_id=100) == n and ix.len() == n and ix.dim() == d
    q = rng.random((5, d), dtype=np.float32)
    check_against_oracle(vdb, 0, rows, q, 10, ids=np.arange(100, 100 + n, dtype=np.uint64), ix=ix)
    with open(path, "r+b") as f:                     # truncated body -> error, nothing loaded
        f.truncate(8 + (n - 1) * d * 4)
    ix2 = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean)
    with pytest.raises(vdb.VectorDbError):
        ix2.load_vector_file(str(path))
    assert ix2.len() == 0


def test_vector_file_header_that_overflows_the_size_check(vdb, tmp_path):
    """`8 + count*dim*4` wraps a size_t for a crafted header (dim = count = 2^31 -> product 2^64 = 0), the
    truncation check passed on an 8-byte file and the loader read far past the mapping.  Also dim > 16384 is refused."""
    for dim, count in ((2 ** 31, 2 ** 31), (2 ** 30, 2 ** 32 - 1), (16385, 1), (4096, 2 ** 32 - 1)):
        path = tmp_path / f"bad_{dim}_{count}.bin"
        with open(path, "wb") as f:
            f.write(np.array([dim, count], dtype="<u4").tobytes())
            f.write(b"\0" * 64)
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean)
        with pytest.raises(vdb.VectorDbError):
            ix.load_vector_file(str(path))
        assert ix.len() == 0


def test_duplicate_ids_inside_one_device_batch_are_last_wins(vdb):
    """vdb_flat_add_bulk_device kept BOTH rows of an id that occurs twice in one batch (len over-counted, the id
    returned twice).  HashMap::insert is last-wins (flat_index.rs:38-41), like the host path."""
    import torch
    rng = np.random.default_rng(5)
    n, d = 3000, 24
    rows = rng.random((n, d), dtype=np.float32)
    ids = np.arange(n, dtype=np.uint64)
    ids[100] = 7          # id 7 twice: rows 7 and 100 -> row 100 wins
    ids[2000] = 7         # ... and a third time: row 2000 wins
    ids[2999] = 1500      # id 1500: row 2999 wins
    t = torch.from_numpy(rows).cuda()
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, keep_host_copy=False)
    ix.add_bulk_device(t.data_ptr(), n, d, ids=ids)
    assert ix.len() == n - 3
    live = np.ones(n, dtype=np.uint8)
    live[[7, 100, 1500]] = 0
    q = np.concatenate([rows[[7, 100, 2000, 1500, 2999]], rng.random((3, d), dtype=np.float32)])
    check_against_oracle(vdb, 0, rows, q, 10, ids=ids, ix=ix, live=live)
    assert np.array_equal(ix.get_vector(7).data, rows[2000]) and np.array_equal(ix.get_vector(1500).data, rows[2999])
    ix.remove(7)
    assert ix.len() == n - 4 and ix.get_vector(7) is None
    # a second batch that overwrites ids of the first one AND repeats an id inside itself
    rows2 = rng.random((4, d), dtype=np.float32)
    t2 = torch.from_numpy(rows2).cuda()
    ix.add_bulk_device(t2.data_ptr(), 4, d, ids=np.array([5, 9, 5, 10**9], dtype=np.uint64))
    assert ix.len() == n - 4 - 2 + 3 and np.array_equal(ix.get_vector(5).data, rows2[2])


def test_concurrent_searches_from_several_threads(vdb):
    """search may be called concurrently on one handle (the server holds RwLock::read() around it,
    src/server/routes.rs:244,:342); ctypes releases the GIL, the library serialises device submission."""
    import threading
    rng = np.random.default_rng(23)
    rows = rng.random((30000, 24), dtype=np.float32)
    ix = make_index(vdb, 0, rows)
    qs = [rng.random((11, 24), dtype=np.float32) for _ in range(6)]
    want = [ix.search_batch_arrays(q, 5) for q in qs]
    got = [None] * 6

    def work(i):
        for _ in range(5):
            got[i] = ix.search_batch_arrays(qs[i], 5)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for w, g in zip(want, got):
        assert all(np.array_equal(a, b) for a, b in zip(w, g))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_candidate_list_distances_match_reference(vdb, metric):
    """BASELINE config 5's offload unit: distances of a query to explicit candidate lists (the calls an
    HNSW search_layer makes, src/hnsw/graph.rs:155,:182), bit-identical to DistanceMetric::distance."""
    rng = np.random.default_rng(31 + metric)
    n, d = 5000, 77
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ids = rng.permutation(np.arange(100, 100 + n)).astype(np.uint64)
    ix = make_index(vdb, metric, rows, ids)
    queries = rng.standard_normal((9, d)).astype(np.float32)
    lists = [rng.choice(ids, size=rng.integers(0, 33), replace=False) for _ in range(9)]   # <= 2m = 32 per expansion
    lists[3] = np.array([ids[0], np.uint64(10 ** 12)], dtype=np.uint64)                 # an id that is not stored
    got = ix.distances_batch(queries, lists)
    row_of = {int(v): i for i, v in enumerate(ids)}
    for b, (l, g) in enumerate(zip(lists, got)):
        assert len(g) == len(l)
        for j, vid in enumerate(l):
            if int(vid) in row_of:
                want = oracle.distance(metric, queries[b], rows[row_of[int(vid)]])
                assert np.float32(g[j]).view(np.uint32) == np.float32(want).view(np.uint32)
            else:
                assert np.isnan(g[j])
    if metric == 1:
        ix.add(7, vdb.Vector(np.zeros(d, np.float32)))
        with pytest.raises(vdb.InvalidVector):
            ix.distances_batch(queries[:1], [[7]])


@pytest.mark.parametrize("n,d,nq,k,metric", [
    (16384, 32, 33, 10, 0),     # largest index still on the dense path
    (16385, 32, 33, 10, 1),     # smallest index on the fused path: 2 blocks of 32 rows per workgroup
    (16511, 1, 5, 3, 2),        # dimension 1 (K padded 1 -> 32)
    (17000, 2, 129, 10, 0),     # 129 queries: 8-wave shape with 127 padding queries
    (17000, 31, 255, 1, 1),
    (20000, 33, 257, 10, 2),    # two passes: 256 queries + 1 query (32-query shape)
    (24000, 100, 64, 10, 0),    # exactly the 64-query shape
    (24000, 100, 65, 10, 1),    # 128-query shape with padding
    (33000, 7, 128, 32, 2),     # k = 32 -> kp = 64
    (40000, 16, 31, 26, 0),     # k + slack == 32 boundary
    (40000, 16, 31, 27, 0),     # first k that needs kp = 64
])
def test_shape_boundaries(vdb, n, d, nq, k, metric):
    rng = np.random.default_rng(n + 13 * d + nq)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = check_against_oracle(vdb, metric, rows, q, k, qsel=sorted({0, nq // 2, nq - 1}))
    st = ix.last_stats()
    assert st["pool_overflows"] == 0, st
    assert (st["rows_scanned"] > 0) == (n > 16384), st


def test_two_searches_in_flight_submit_wait(vdb):
    """vdb_flat_search_batch_device_submit / _wait: two batches in flight on two contexts and streams.  Results equal the
    synchronous call's bit for bit whatever the interleaving; a third submit, a mutation or a host-pointer search while a
    ticket is outstanding is refused; errors surface at wait like at the synchronous call; the handle stays usable."""
    import torch
    rng = np.random.default_rng(77)
    n, d, B, k = 120_000, 64, 200, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    ix = make_index(vdb, 1, rows)
    qs = [torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32)).to(dev) for _ in range(5)]
    outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(5)]
    ref = []
    for q in qs:                                                # synchronous reference
        o = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty((B,), dtype=torch.int32, device=dev))
        ix.search_batch_device(q.data_ptr(), B, d, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        ref.append(tuple(t.clone() for t in o))
    torch.cuda.synchronize()

    def submit(i):
        return ix.search_batch_device_submit(qs[i].data_ptr(), B, d, k, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr())

    t0 = submit(0)
    for i in range(1, 5):                                       # keep two in flight
        t1 = submit(i)
        with pytest.raises(vdb.VectorDbError):
            submit(0)                                           # a third ticket is refused
        with pytest.raises(vdb.VectorDbError):
            ix.add(10**7, vdb.Vector(rows[0]))                  # the row store must not change under a search in flight
        with pytest.raises(vdb.VectorDbError):
            ix.search_batch_arrays(rows[:2], 3)                 # host-pointer entry: refused as well
        ix.search_batch_device_wait(t0)
        t0 = t1
    ix.search_batch_device_wait(t0)
    with pytest.raises(vdb.VectorDbError):
        ix.search_batch_device_wait(t0)                         # a ticket is waited for exactly once
    torch.cuda.synchronize()
    for i in range(5):
        for a, b in zip(outs[i], ref[i]):
            assert torch.equal(a, b), i
    st = ix.last_stats()
    assert st["bf16_screen"] == 1 and st["mfma_queries"] == B
    # a synchronous device search may run beside ONE outstanding ticket
    t = submit(1)
    o = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
         torch.empty((B,), dtype=torch.int32, device=dev))
    ix.search_batch_device(qs[2].data_ptr(), B, d, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    ix.search_batch_device_wait(t)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(o, ref[2])) and all(torch.equal(a, b) for a, b in zip(outs[1], ref[1]))
    # an error found on the device (zero-norm query under Cosine) comes back from wait; the slot is free again afterwards
    qz = qs[0].clone()
    qz[7] = 0.0
    t = ix.search_batch_device_submit(qz.data_ptr(), B, d, k, outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr())
    with pytest.raises(vdb.InvalidVector):
        ix.search_batch_device_wait(t)
    t = submit(3)
    ix.search_batch_device_wait(t)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs[3], ref[3]))
    ix.add(10**7, vdb.Vector(rows[0]))                          # mutations work again
    assert ix.len() == n + 1


def test_sample_cache_with_two_searches_of_different_sample_size_in_flight(vdb):
    """The compact bf16 copy of the sample rows is keyed by (rows uploaded, sample size).  Two searches in flight whose plans
    differ in sample size (k = 10 and k = 100 on a 100k-row index: 16384 and 32768 sample rows) cannot both use it: the second
    one finds the other workspace busy and keeps the f32 gather for its sample pass.  Results are the synchronous ones either
    way, in both submission orders, and the copy is rebuilt for whoever comes next."""
    import torch
    rng = np.random.default_rng(123)
    n, d, B = 100_000, 128, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    ix = make_index(vdb, 1, rows)
    q = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32)).to(dev)

    def bufs(k):
        return (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
                torch.empty((B,), dtype=torch.int32, device=dev))

    ref = {}
    for k in (10, 100):
        o = bufs(k)
        ix.search_batch_device(q.data_ptr(), B, d, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        torch.cuda.synchronize()
        ref[k] = tuple(t.clone() for t in o)
        assert ix.last_stats()["bf16_screen"] == 1
    assert ix.last_stats()["sample_rows"] == 32768                  # k = 100 doubled the sample
    for order in ((10, 100), (100, 10), (10, 100)):
        outs = {k: bufs(k) for k in order}
        tickets = [ix.search_batch_device_submit(q.data_ptr(), B, d, k, outs[k][0].data_ptr(), outs[k][1].data_ptr(), outs[k][2].data_ptr())
                   for k in order]
        for t in tickets:
            ix.search_batch_device_wait(t)
        torch.cuda.synchronize()
        for k in order:
            assert all(torch.equal(a, b) for a, b in zip(outs[k], ref[k])), (order, k)
    qh = q.cpu().numpy()
    gi, gd, gc = ix.search_batch_arrays(qh, 10)
    oi, od = oracle.flat_search(1, rows, qh[3], 10)
    assert np.array_equal(gi[3], oi) and np.array_equal(gd[3].view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_direct_path_of_small_indexes_equals_the_tiered_pipeline_and_the_oracle(vdb, metric):
    """Indexes of at most 16384 rows, batches of at most 8 queries: two kernels (exact scan of every row in the reference's
    operation order, select + emit) through mapped host memory.  Same ids, order, counts and distance bits as the tiered
    pipeline (VDB_TIERS_NO_DIRECT) and as the oracle -- with tombstones, sparse non-monotone ids, an id mask, per-query k, k > n,
    exact ties, and the reference's error semantics."""
    rng = np.random.default_rng(40 + metric)
    n, d = 9000, 70
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[100] = rows[7000]                                             # an exact tie, decided by id
    ids = rng.permutation(np.arange(1, 40 * n, 37, dtype=np.uint64))[:n]      # sparse, not monotone
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    ix.add_bulk(rows, ids=ids)
    dead = np.zeros(n, dtype=bool); dead[::13] = True
    for i in ids[dead]:
        ix.remove(int(i))
    live = (~dead).astype(np.uint8)
    q = rng.standard_normal((8, d)).astype(np.float32)
    q[3] = rows[100]
    ks = np.array([1, 10, 64, 3, 2048, 7, 100, 5000], dtype=np.uintp)   # 5000 > 2048: that batch takes the tiered pipeline by itself
    for qsel, kk in ((slice(0, 7), ks[:7]), (slice(0, 1), 10), (slice(0, 8), 10), (slice(0, 8), ks)):
        qq = q[qsel]
        a = ix.search_batch_arrays(qq, kk)
        st = ix.last_stats()
        kmax = int(np.max(kk))
        assert st["exact_queries"] == (len(qq) if kmax <= 2048 else st["exact_queries"]), st
        ix.set_tiers(ix.TIERS_NO_DIRECT)
        b = ix.search_batch_arrays(qq, kk)
        ix.set_tiers(0)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        for bq in range(len(qq)):
            kb = int(kk if np.isscalar(kk) else kk[bq])
            oi, od = oracle.flat_search(metric, rows, qq[bq], kb, ids=ids, live=live)
            assert a[2][bq] == len(oi) and np.array_equal(a[0][bq, :len(oi)], oi) and np.array_equal(a[1][bq, :len(od)].view(np.uint32), od.view(np.uint32))
    # id mask (pre-filter) through the direct path: host-pointer entry point falls back to its device staging, the path is the same
    mlive = (rng.random(int(ids.max()) + 1) < 0.2)
    mask = np.packbits(mlive.astype(np.uint8), bitorder="little")
    mask = np.concatenate([mask, np.zeros((-len(mask)) % 8, dtype=np.uint8)]).view(np.uint64)
    m = ix.search_batch_arrays(q[:4], 10, id_mask=mask, mask_bits=len(mlive))
    assert ix.last_stats()["exact_queries"] == 4
    lv = live & mlive[ids.astype(np.int64)].astype(np.uint8)
    for bq in range(4):
        oi, od = oracle.flat_search(metric, rows, q[bq], 10, ids=ids, live=lv)
        assert m[2][bq] == len(oi) and np.array_equal(m[0][bq, :len(oi)], oi) and np.array_equal(m[1][bq, :len(od)].view(np.uint32), od.view(np.uint32))
    # errors: dimension mismatch, zero-norm query and a zero-norm row under Cosine, NaN
    with pytest.raises(vdb.DimensionMismatch):
        ix.search_batch_arrays(np.ones((1, d + 3), dtype=np.float32), 5)
    if metric == 1:
        with pytest.raises(vdb.InvalidVector):
            ix.search_batch_arrays(np.zeros((2, d), dtype=np.float32), 5)
        ix.add(10**9, vdb.Vector(np.zeros(d, dtype=np.float32)))
        with pytest.raises(vdb.InvalidVector):
            ix.search_batch_arrays(q[:2], 5)
        ix.remove(10**9)
    qn = q[:2].copy(); qn[1, 5] = np.nan
    with pytest.raises(vdb.NanDistance):
        ix.search_batch_arrays(qn, 5)
    again = ix.search_batch_arrays(q[:7], ks[:7])                        # the handle stays usable, the status word was cleaned
    first = ix.search_batch_arrays(q[:7], ks[:7])
    for x, y in zip(again, first):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
