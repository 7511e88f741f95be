"""The bf16 screening tier (kernels_fused_bf16.hip) against the CPU oracle and against the f32 MFMA tier.

The tier only RANKS rows on the bf16 matrix cores; what it returns are the reference's exact f32 distances
(distance.rs:37-73) of certified candidates, so everything here is compared with tolerance 0: ids, order,
counts and the bit patterns of the distances."""
import os

import numpy as np
import pytest

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu


_SHADOW = [False]


@pytest.fixture(scope="module", params=["f32_rows", "bf16_shadow"])
def vdb(request):
    """Every test of this module runs twice: screening from the f32 rows (kernels_fused_bf16p.hip) and from the opt-in bf16
    shadow copy (kernels_fused_s16.hip, vdb_flat_set_shadow on every index the tests build).  Expectations are the same --
    the shadow changes which bytes the filter pass streams, never a result or a tier counter."""
    v = load_package()
    v.build()
    _SHADOW[0] = request.param == "bf16_shadow"
    yield v
    _SHADOW[0] = False


def shadow_on():
    return _SHADOW[0]


def make_index(vdb, metric, rows, ids=None):
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    if shadow_on():
        ix.set_shadow(True)                                       # before the first add: the store grows with its shadow
    ix.add_bulk(rows, ids=ids)
    return ix


def same(a, b):
    return all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))


def both_tiers(ix, q, k, **kw):
    """(results with the screening tier, its stats, results with the f32 tier only)"""
    ix.set_screen(1)
    a = ix.search_batch_arrays(q, k, **kw)
    st = ix.last_stats()
    ix.set_screen(0)
    b = ix.search_batch_arrays(q, k, **kw)
    st0 = ix.last_stats()
    ix.set_screen(1)
    assert st0["bf16_screen"] == 0
    return a, st, b


def check_oracle(metric, rows, q, k, res, qsel, ids=None, live=None):
    gi, gd, gc = res
    for b in qsel:
        oi, od = oracle.flat_search(metric, rows, q[b], k, ids=ids, live=live)
        assert gc[b] == len(oi), (b, gc[b], len(oi))
        assert np.array_equal(gi[b, :gc[b]], oi), (b, gi[b, :gc[b]], oi)
        assert np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)), (b, gd[b, :gc[b]], od)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq,k,dist", [
    (65536, 32, 40, 10, "gauss"),        # smallest index the tier takes; one K stage per tile
    (65537, 64, 256, 1, "uniform"),      # two K stages; k = 1
    (100001, 100, 33, 10, "uniform"),    # dimension padded 100 -> 128, ragged last tile
    (131072, 768, 256, 10, "uniform"),   # the headline dimension
    (90000, 1536, 70, 16, "gauss"),      # ada-002 dimension
    (200000, 48, 300, 50, "gauss"),      # two passes (256 + 44 queries), k = 50
    (150000, 96, 17, 112, "uniform"),    # largest k the tier serves
])
def test_screen_tier_vs_oracle_and_f32_tier(vdb, metric, n, d, nq, k, dist):
    rng = np.random.default_rng(n + 3 * d + nq + metric)
    if dist == "uniform":
        rows = rng.random((n, d), dtype=np.float32)
        q = rng.random((nq, d), dtype=np.float32)
    else:
        rows = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = make_index(vdb, metric, rows)
    a, st, b = both_tiers(ix, q, k)
    assert st["bf16_screen"] == 1 and st["kprime"] == (512 if k > 48 else 256) and st["rows_scanned"] >= n, st
    assert st["pool_overflows"] == 0, st
    assert st["shadow_rows"] == (1 if shadow_on() and ((d + 31) // 32) % 2 == 0 else 0), st
    assert same(a, b)
    check_oracle(metric, rows, q, k, a, sorted({0, nq // 2, nq - 1}))


def test_screen_tier_not_used_below_its_minimum_or_above_its_k(vdb):
    rng = np.random.default_rng(3)
    rows = rng.random((65535, 16), dtype=np.float32)
    q = rng.random((4, 16), dtype=np.float32)
    ix = make_index(vdb, 0, rows)
    ix.search_batch_arrays(q, 10)
    assert ix.last_stats()["bf16_screen"] == 0
    rows = rng.random((70000, 16), dtype=np.float32)
    ix = make_index(vdb, 0, rows)
    r = ix.search_batch_arrays(q, 113)                                   # k > 112: the f32 tier (kp = 0 -> exact scan)
    assert ix.last_stats()["bf16_screen"] == 0
    check_oracle(0, rows, q, 113, r, [0, 3])
    ix.search_batch_arrays(q, 112)
    assert ix.last_stats()["bf16_screen"] == 1


def test_uncertified_queries_go_through_the_f32_tier(vdb):
    """VDB_TIERS_FORCE_F32 (vdb_flat_set_tiers) sends every query of the screening tier to the compact f32 re-run: same results."""
    rng = np.random.default_rng(21)
    rows = rng.standard_normal((80000, 72)).astype(np.float32)
    q = rng.standard_normal((37, 72)).astype(np.float32)
    for metric in (0, 1, 2):
        ix = make_index(vdb, metric, rows)
        a = ix.search_batch_arrays(q, 10)
        assert ix.last_stats()["f32_tier_queries"] == 0
        ix.set_tiers(ix.TIERS_FORCE_F32)
        b = ix.search_batch_arrays(q, 10)
        st = ix.last_stats()
        ix.set_tiers(0)
        assert st["bf16_screen"] == 1 and st["f32_tier_queries"] == 37, st
        assert same(a, b)
        check_oracle(metric, rows, q, 10, b, [0, 36])


def test_duplicates_and_near_ties_fall_through_the_tiers(vdb):
    """Every row 160 times: the k-th and (k+1)-th distances tie exactly, nothing can be certified by a score
    bound, and the answer must still be the oracle's (distance, then id)."""
    rng = np.random.default_rng(22)
    base = rng.random((500, 40), dtype=np.float32)
    rows = np.concatenate([base] * 160, 0)                                # 80000 rows
    ids = rng.permutation(rows.shape[0]).astype(np.uint64)
    q = base[:5] + 0.0
    for metric in (0, 1, 2):
        ix = make_index(vdb, metric, rows, ids=ids)
        a, st, b = both_tiers(ix, q, 10)
        assert st["bf16_screen"] == 1
        assert same(a, b)
        check_oracle(metric, rows, q, 10, a, range(5), ids=ids)


def test_near_duplicates_within_bf16_resolution(vdb):
    """Rows that differ by less than bf16 can resolve (relative 1e-4): the screening scores cannot order them,
    the exact re-rank must."""
    rng = np.random.default_rng(23)
    n, d = 70000, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    rows[1000:1300] = centre + 1e-4 * rng.standard_normal((300, d)).astype(np.float32)
    q = np.stack([centre, centre + 1e-5, rows[1100]]).astype(np.float32)
    for metric in (0, 1, 2):
        ix = make_index(vdb, metric, rows)
        a, st, b = both_tiers(ix, q, 10)
        assert same(a, b)
        check_oracle(metric, rows, q, 10, a, range(3))


def test_rethreshold_pass_answers_what_the_depth_limit_could_not(vdb):
    """600 rows inside the bf16 error bound of each other: the first pass re-ranks 256 candidates without a certificate,
    but its k-th exact distance implies a score cut; the re-threshold pass filters with that cut, re-ranks EVERY key
    under it and is exact by construction -- no f32 tier, no exact scan."""
    rng = np.random.default_rng(28)
    n, d = 120000, 96
    rows = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32) * 2.0
    rows[40000:40600] = centre + 1e-3 * rng.standard_normal((600, d)).astype(np.float32)
    q = (centre + 1e-4 * rng.standard_normal((5, d))).astype(np.float32)
    for metric in (0, 1, 2):
        ix = make_index(vdb, metric, rows)
        a, st, b = both_tiers(ix, q, 10)
        assert st["bf16_screen"] == 1 and st["rethreshold_queries"] == 5 and st["f32_tier_queries"] == 0 and st["exact_queries"] == 0, st
        assert same(a, b)
        check_oracle(metric, rows, q, 10, a, range(5))


def test_clustered_row_order_stays_on_the_screen_tier(vdb):
    """All near neighbours sit in ONE row range (data stored cluster by cluster): nearly every key that passes the
    threshold lands in one workgroup's private sub-pools.  They are sized for that (4 x 256 slots per query and
    workgroup), so the query is still answered -- and certified -- by the screening tier."""
    rng = np.random.default_rng(24)
    n, d = 200000, 32
    rows = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    rows[5000:5300] = centre + 0.3 * rng.standard_normal((300, d)).astype(np.float32)  # a cluster of 300 rows, stored together
    q = (centre + 0.05 * rng.standard_normal((3, d))).astype(np.float32)
    ix = make_index(vdb, 0, rows)
    a, st, b = both_tiers(ix, q, 10)
    assert st["bf16_screen"] == 1 and st["pool_overflows"] == 0 and st["exact_queries"] == 0, st
    assert same(a, b)
    check_oracle(0, rows, q, 10, a, range(3))


def test_pool_overflow_is_routed_to_the_next_tiers(vdb):
    """More near-duplicates in one tile than a sub-pool can hold (2000 copies of one row): the overflow flag must send
    the query on, and the answer must still be the oracle's (distance, then id)."""
    rng = np.random.default_rng(27)
    n, d = 100000, 16
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[30000:32000] = rows[30000]                                                    # 2000 identical rows
    q = np.stack([rows[30000], rows[30000] + 1e-3]).astype(np.float32)
    ix = make_index(vdb, 0, rows)
    a, st, b = both_tiers(ix, q, 10)
    assert same(a, b)
    check_oracle(0, rows, q, 10, a, range(2))


def test_tombstones_overwrites_and_prefilter_under_the_screen_tier(vdb):
    rng = np.random.default_rng(25)
    n, d = 90000, 24
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((20, d), dtype=np.float32)
    ix = make_index(vdb, 1, rows)
    live = np.ones(n, dtype=np.uint8)
    for i in range(0, n, 7):
        ix.remove(i)
        live[i] = 0
    a, st, b = both_tiers(ix, q, 10)
    assert st["bf16_screen"] == 1
    assert same(a, b)
    check_oracle(1, rows, q, 10, a, [0, 19], live=live)
    # pre-filter: ids with id % 5 == 0 only (device bitmask), on top of the tombstones
    keep = (np.arange(n) % 5 == 0)
    mask = np.zeros((n + 63) // 64, dtype=np.uint64)
    idx = np.nonzero(keep)[0]
    np.bitwise_or.at(mask, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
    ix.set_screen(1)
    r = ix.search_batch_arrays(q, 10, id_mask=mask, mask_bits=n)
    assert ix.last_stats()["bf16_screen"] == 1
    check_oracle(1, rows, q, 10, r, [0, 7, 19], live=(live & keep.astype(np.uint8)))
    # a very selective filter: fewer eligible rows than the threshold rank -> every eligible row is a candidate
    keep2 = (np.arange(n) % 3001 == 1)
    mask2 = np.zeros_like(mask)
    idx = np.nonzero(keep2)[0]
    np.bitwise_or.at(mask2, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
    r = ix.search_batch_arrays(q, 10, id_mask=mask2, mask_bits=n)
    check_oracle(1, rows, q, 10, r, [0, 19], live=(live & keep2.astype(np.uint8)))


def test_error_semantics_under_the_screen_tier(vdb):
    V = vdb.Vector
    rng = np.random.default_rng(26)
    rows = rng.random((70000, 8), dtype=np.float32)
    ix = make_index(vdb, 1, rows)
    with pytest.raises(vdb.InvalidVector):                                # zero-norm query under Cosine (distance.rs:51-55)
        ix.search(V([0.0] * 8), 3)
    with pytest.raises(vdb.DimensionMismatch):
        ix.search(V([1.0] * 7), 3)
    ix.add(10**6, V([0.0] * 8))                                           # ONE zero-norm row fails every search (flat_index.rs:57-60)
    with pytest.raises(vdb.InvalidVector):
        ix.search(V([1.0] * 8), 3)
    ix.remove(10**6)
    assert len(ix.search(V([1.0] * 8), 3)) == 3
    e = make_index(vdb, 0, rows)
    e.add(10**6, V([float("nan")] + [1.0] * 7))                           # NaN distance: the reference panics (flat_index.rs:62)
    with pytest.raises(vdb.VectorDbError):
        e.search(V([1.0] * 8), 3)


def test_shadow_copy_follows_adds_growth_and_toggling(vdb):
    """vdb_flat_set_shadow on a filled index converts the existing rows; later adds (including a reallocation of the
    store) maintain the shadow; turning it off frees it.  The results never change."""
    rng = np.random.default_rng(99)
    n0, n1, d, nq, k = 70_000, 90_000, 100, 50, 10
    rows = rng.standard_normal((n0 + n1, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(1), keep_host_copy=False)
    ix.add_bulk(rows[:n0])
    base = ix.search_batch_arrays(q, k)
    assert ix.last_stats()["shadow_rows"] == 0
    ix.set_shadow(True)
    r = ix.search_batch_arrays(q, k)
    assert ix.last_stats()["shadow_rows"] == 1 and ix.last_stats()["bf16_screen"] == 1 and same(r, base)
    ix.add_bulk(rows[n0:], first_id=n0)                       # grows the store: the shadow is reallocated and extended
    r2 = ix.search_batch_arrays(q, k)
    st = ix.last_stats()
    assert st["shadow_rows"] == 1 and st["rows_scanned"] >= n0 + n1
    check_oracle(1, rows, q, k, r2, [0, 17, nq - 1])
    for j in range(0, 40):                                    # single adds through the host staging path
        ix.add(10_000_000 + j, vdb.Vector(rows[j] * 1.5))
    ix.remove(5)
    r3 = ix.search_batch_arrays(q, k)
    assert ix.last_stats()["shadow_rows"] == 1
    ix.set_shadow(False)
    r4 = ix.search_batch_arrays(q, k)
    assert ix.last_stats()["shadow_rows"] == 0 and same(r3, r4)
    ix.set_screen(0)
    assert same(ix.search_batch_arrays(q, k), r4)


def test_shadow_and_f32_rows_give_the_same_raw_screening_scores(vdb):
    """The scores themselves, not only the results: every (query, row) key the production filter pass emits carries the same
    f32 bits from the bf16 shadow as from the f32 rows (same roundings, same MFMA order per accumulator), for the plain
    scores and for the lower-bound scores of Dot / Euclid, with a ragged last tile and tombstones."""
    rng = np.random.default_rng(5)
    n, d, nq = 66_000 + 77, 192, 19
    rows = (rng.standard_normal((n, d)) * rng.uniform(0.1, 4.0, (n, 1))).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    for metric in (0, 1, 2):
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows)
        for r in range(0, 2000, 3):
            ix.remove(r)
        for raw in (0, 1):                                        # the scores the tier ranks by (lower bounds under Dot / Euclid), the plain scores
            a = ix.debug_screen_scores(q, raw=raw)
            ix.set_shadow(True)
            b = ix.debug_screen_scores(q, raw=raw)
            ix.set_shadow(False)
            sa, sb = a[0], b[0]
            assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), (metric, raw)


def test_sample_cache_gives_the_same_thresholds_and_results(vdb):
    """The compact bf16 copy of the sample rows (default) against the f32 gather (vdb_flat_set_sample_cache(0)): the per-query
    filter thresholds are the same f32 bits -- same roundings, same MFMA order, same group minima -- and so is everything
    downstream; the copy follows adds (rebuilt by the next search), tombstones and an id mask."""
    rng = np.random.default_rng(31)
    n0, n1, d, nq, k = 100_000, 37_777, 128, 200, 10
    rows = (rng.standard_normal((n0 + n1, d)) * rng.uniform(0.2, 3.0, (n0 + n1, 1))).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    for metric in (0, 1, 2):
        ix = make_index(vdb, metric, rows[:n0])
        for r in range(0, 5000, 11):
            ix.remove(r)

        def both(**kw):
            ix.set_sample_cache(True)
            a = ix.search_batch_arrays(q, k, **kw)
            ta, sa = ix.debug_last_thresholds(nq), ix.last_stats()
            ix.set_sample_cache(False)
            b = ix.search_batch_arrays(q, k, **kw)
            tb, sb = ix.debug_last_thresholds(nq), ix.last_stats()
            ix.set_sample_cache(True)
            assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32)), metric
            assert np.all(np.isfinite(ta))
            assert same(a, b)
            for key in ("bf16_screen", "uncertified", "pool_overflows", "f32_tier_queries", "exact_queries", "rethreshold_queries", "sample_rows"):
                assert sa[key] == sb[key], (key, sa[key], sb[key])
            assert sa["bf16_screen"] == 1
            return a

        both()
        ix.add_bulk(rows[n0:], first_id=n0)                       # more rows: other sample rows, the copy is rebuilt
        r2 = both()
        live = np.ones(n0 + n1, dtype=np.uint8)
        live[0:5000:11] = 0
        check_oracle(metric, rows, q, k, r2, [0, nq - 1], live=live)
        mask = np.zeros(((n0 + n1 + 63) // 64,), dtype=np.uint64)  # an id mask: every third id
        ids = np.arange(0, n0 + n1, 3)
        np.bitwise_or.at(mask, ids // 64, np.uint64(1) << (ids % 64).astype(np.uint64))
        both(id_mask=mask, mask_bits=n0 + n1)


def test_nan_and_wild_norms_under_the_min_test_of_the_epilogue(vdb):
    """The filter epilogues test the MINIMUM of four scores against the threshold when no score of the launch can be NaN
    (every row and query norm within [2^-40, 2^40]: kernels.h fused_no_nan) and add a NaN test per group otherwise.  A NaN row
    must still reach the exact re-rank (the reference panics on a NaN distance, flat_index.rs:62 -> VDB_ERR_NAN), and rows /
    queries outside the tame domain must not change a result."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 70_000, 64, 24, 10
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    for metric in (0, 1, 2):
        # (a) a NaN element somewhere in the index: every search fails with the NaN error.  An inf element gives inf (Euclid, Dot:
        # the reference sorts it without complaint) or NaN (Cosine) distances: whatever the f32 tier does, the screening tier does
        for bad in (float("nan"), float("inf")):
            rows = base.copy()
            rows[12345, 3] = bad
            ix = make_index(vdb, metric, rows)
            outcome = []
            for screen in (1, 0):
                ix.set_screen(screen)
                try:
                    outcome.append(ix.search_batch_arrays(q, k))
                except vdb.VectorDbError as e:
                    outcome.append(str(e))
                assert ix.last_stats()["bf16_screen"] == screen
            ix.set_screen(1)
            if bad != bad or metric == 1:
                assert isinstance(outcome[0], str) and "NaN" in outcome[0] and outcome[0] == outcome[1], outcome
            else:
                assert not isinstance(outcome[0], str) and same(outcome[0], outcome[1])
                check_oracle(metric, rows, q, k, outcome[0], [0, nq - 1])
        # (b) a row and (c) a query far outside the tame domain, (d) rows with tiny norms: results equal the f32 tier's and the oracle's
        rows = base.copy()
        rows[777] *= np.float32(1e25)
        rows[778] *= np.float32(1e-22)
        ix = make_index(vdb, metric, rows)
        a, st, b = both_tiers(ix, q, k)
        assert st["bf16_screen"] == 1 and same(a, b)
        check_oracle(metric, rows, q, k, a, [0, nq - 1])
        q2 = q.copy()
        q2[5] *= np.float32(1e18)
        ix2 = make_index(vdb, metric, base)
        a2, st2, b2 = both_tiers(ix2, q2, k)
        assert st2["bf16_screen"] == 1 and same(a2, b2)
        check_oracle(metric, base, q2, k, a2, [5, 6])
        # and the next, tame search on the same handle is not stuck with the wild query's norm
        a3, st3, b3 = both_tiers(ix2, q, k)
        assert same(a3, b3)
        check_oracle(metric, base, q, k, a3, [0])


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq,k", [
    (70001, 64, 257, 10),        # one wide pass whose second block holds ONE query; ragged last 128-row tile
    (140000, 96, 512, 10),       # exactly one wide pass
    (200000, 48, 700, 50),       # a wide pass (512) + an ordinary pass (188 queries), k' = 512
    (131072, 768, 1024, 100),    # BASELINE config 3's batch and k at the headline dimension: two wide passes
    (90000, 100, 300, 16),       # dimension padded 100 -> 128: the shadow rows are not usable, the wide kernel is
])
def test_wide_filter_pass_equals_the_256_query_passes(vdb, metric, n, d, nq, k):
    """Batches above 256 queries: the 128-row x 512-query kernel (kernels_fused_bf16w.hip) serves two 256-query blocks per
    fetch of the rows.  Same thresholds, same pools' content, same results, bit for bit, as the 256-query passes and as the
    f32 tier; the rows are scanned ceil(B / 512) times instead of ceil(B / 256)."""
    rng = np.random.default_rng(5 * n + d + nq + metric)
    rows = rng.random((n, d), dtype=np.float32)
    q = rng.random((nq, d), dtype=np.float32)
    ix = make_index(vdb, metric, rows)
    for i in range(0, n, 997):                                    # tombstones in every tile neighbourhood
        ix.remove(i)
    live = np.ones(n, dtype=np.uint8); live[::997] = 0
    ix.set_wide(True)
    a = ix.search_batch_arrays(q, k)
    st_w = ix.last_stats()
    thr_w = ix.debug_last_thresholds(nq)
    ix.set_wide(False)
    b = ix.search_batch_arrays(q, k)
    st_n = ix.last_stats()
    thr_n = ix.debug_last_thresholds(nq)
    ix.set_screen(0)
    c = ix.search_batch_arrays(q, k)
    ix.set_screen(1); ix.set_wide(True)
    assert same(a, b) and same(a, c)
    assert np.array_equal(thr_w.view(np.uint32), thr_n.view(np.uint32))
    wide_used = not (shadow_on() and ((d + 31) // 32) % 2 == 0)    # the shadow rows' kernel has the one shape
    assert st_n["rows_scanned"] == -(-nq // 256) * n, st_n
    assert st_w["rows_scanned"] == (-(-nq // 512) * n if wide_used else st_n["rows_scanned"]), st_w
    for key in ("mfma_queries", "exact_queries", "pool_overflows", "uncertified", "f32_tier_queries", "rethreshold_queries"):
        assert st_w[key] == st_n[key], (key, st_w, st_n)
    assert st_w["pool_overflows"] == 0 and st_w["exact_queries"] == 0, st_w
    check_oracle(metric, rows, q, k, a, sorted({0, 255, 256, nq // 2, nq - 1}), live=live)


def test_wide_filter_pass_with_prefilter_mask_and_clustered_rows(vdb):
    """The wide kernel's rare path: an id mask at 3 % selectivity and a cluster stored contiguously (its keys land in ONE
    workgroup's two sub-pools per query -- this shape has two per workgroup where the 256-query kernel has four)."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 120000, 64, 600, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    rows[50000:50400] = centre + 0.05 * rng.standard_normal((400, d)).astype(np.float32)      # 400 neighbours in 4 consecutive tiles
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[::7] = centre + 0.05 * rng.standard_normal((len(q[::7]), d)).astype(np.float32)
    ix = make_index(vdb, 0, rows)
    a = ix.search_batch_arrays(q, k)
    st = ix.last_stats()
    ix.set_wide(False)
    b = ix.search_batch_arrays(q, k)
    assert same(a, b) and st["exact_queries"] == ix.last_stats()["exact_queries"]
    ix.set_wide(True)
    check_oracle(0, rows, q, k, a, [0, 7, 300, 301, 599])
    live = (rng.random(n) < 0.03).astype(np.uint8)
    mask = np.packbits(live, bitorder="little")
    mask = np.concatenate([mask, np.zeros((-len(mask)) % 8, dtype=np.uint8)]).view(np.uint64)
    m = ix.search_batch_arrays(q, k, id_mask=mask, mask_bits=n)
    ix.set_wide(False)
    m2 = ix.search_batch_arrays(q, k, id_mask=mask, mask_bits=n)
    assert same(m, m2)
    check_oracle(0, rows, q, k, m, [0, 256, 511, 599], live=live)
