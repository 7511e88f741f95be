"""ONE index over several shards in ONE process (vdb_flat_create_sharded, include/vdb_flat.h; VERDICT r2 row *2): the same
`vdb_flat_*` calls, the same answers as a plain single-GPU index and as the oracle, bit for bit.

The reference's seam is one in-process object (`VectorStore<I: Index>`, src/storage.rs:83,:116-127, held by one server
process, src/server/mod.rs:13-16), so the handle itself owns the row shards.  On the one-GPU box the shards share device 0:
devices=[0] runs the RCCL exchange (a single-rank in-process communicator, ncclCommInitAll), devices=[0,0] / [0,0,0] the
peer-copy exchange (RCCL refuses two ranks on one device) -- every line of the routing, the two-exchange protocol, the merge
and the error paths runs; only the inter-GPU wire is absent."""
import ctypes

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


def _same(a, b):
    return all(np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
               for x, y in zip(a, b))


def _check_oracle(metric, rows, queries, k, got, ids=None, live=None, qsel=None):
    gi, gd, gc = got
    for b in (range(queries.shape[0]) if qsel is None else qsel):
        oi, od = oracle.flat_search(metric, rows, queries[b], k, ids=ids, live=live)
        assert gc[b] == len(oi), (b, gc[b], len(oi))
        assert np.array_equal(gi[b, :gc[b]], oi), (b, gi[b, :gc[b]], oi)
        assert np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)), b


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_sharded_handle_equals_plain_index_and_oracle(vdb, devices, metric):
    rng = np.random.default_rng(10 + metric)
    n, d, B, k = 150_001, 64, 37, 10                      # large enough for the screening tier on every shard (>= 65536 rows for G <= 2)
    rows = rng.random((n, d), dtype=np.float32)
    queries = rng.random((B, d), dtype=np.float32)
    plain = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    plain.add_bulk(rows)
    ref = plain.search_batch_arrays(queries, k)
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), devices=devices, keep_host_copy=False)
    assert sh.shards() == len(devices) and plain.shards() == 1
    sh.add_bulk(rows)
    G = len(devices)
    assert sorted(sh.shard_len(g) for g in range(G)) == sorted([n // G + (1 if g < n % G else 0) for g in range(G)])
    assert sh.len() == n and sh.dim() == d and int(sh.metric()) == metric
    got = sh.search_batch_arrays(queries, k)
    assert _same(ref, got)
    _check_oracle(metric, rows, queries, k, got, qsel=range(0, B, 6))
    st = sh.shard_stats()
    assert st["shards"] == G and st["exchanges"] == 1
    assert st["exchange_mode"] == (sh.EXCHANGE_RCCL if G == 1 else sh.EXCHANGE_PEER)
    assert st["rccl_ranks"] == (1 if G == 1 else 0)
    agg = sh.last_stats()
    assert agg["rows_scanned"] == n and agg["mfma_queries"] == B * G     # every shard answered every query from its first tier


def test_sharded_handle_device_resident_call_and_second_exchange(vdb):
    import torch
    rng = np.random.default_rng(3)
    n, d, B, k = 140_000, 96, 64, 10
    rows = rng.random((n, d), dtype=np.float32)
    queries = rng.random((B, d), dtype=np.float32)
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Cosine, devices=[0, 0], keep_host_copy=False)
    dev = torch.device("cuda", 0)
    r_t = torch.from_numpy(rows).to(dev)
    torch.cuda.synchronize()
    sh.add_bulk_device(r_t.data_ptr(), n, d, first_id=0)               # rows resident on device 0, dealt to the shards on the device
    q_t = torch.from_numpy(queries).to(dev)
    o_i = torch.empty((B, k), dtype=torch.int64, device=dev)
    o_d = torch.empty((B, k), dtype=torch.float32, device=dev)
    o_c = torch.empty((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    sh.search_batch_device(q_t.data_ptr(), B, d, k, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
    first = (o_i.cpu().numpy().astype(np.uint64), o_d.cpu().numpy(), o_c.cpu().numpy())
    _check_oracle(1, rows, queries, k, first, qsel=range(0, B, 9))
    assert sh.shard_stats()["exchanges"] == 1
    # every query handed to the slower tiers on every shard: the shards rewrite their blocks in their second half, the reduced
    # status says so, and the exchange runs a second time -- same answers
    sh.set_tiers(sh.TIERS_FORCE_F32)
    sh.search_batch_device(q_t.data_ptr(), B, d, k, o_i.data_ptr(), o_d.data_ptr(), o_c.data_ptr())
    second = (o_i.cpu().numpy().astype(np.uint64), o_d.cpu().numpy(), o_c.cpu().numpy())
    assert sh.shard_stats()["exchanges"] == 2
    assert _same(first, second)
    assert sh.last_stats()["f32_tier_queries"] == 2 * B
    sh.set_tiers(0)


def test_sharded_handle_mutations_follow_the_reference(vdb):
    """Index::add overwrites (flat_index.rs:38-41), remove of an absent id is Ok (:43-46), get_vector, len; ids stay global."""
    rng = np.random.default_rng(5)
    d = 24
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 0, 0])
    plain = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean)
    rows = {}
    for i in range(300):                                               # single adds: dealt to the emptiest shard
        v = rng.random(d, dtype=np.float32)
        rows[i * 7] = v
        sh.add(i * 7, vdb.Vector(v)); plain.add(i * 7, vdb.Vector(v))
    assert [sh.shard_len(g) for g in range(3)] == [100, 100, 100]
    for i in range(0, 300, 5):                                         # overwrite in place: the id keeps its shard, len is unchanged
        v = rng.random(d, dtype=np.float32)
        rows[i * 7] = v
        sh.add(i * 7, vdb.Vector(v)); plain.add(i * 7, vdb.Vector(v))
    assert sh.len() == 300 and [sh.shard_len(g) for g in range(3)] == [100, 100, 100]
    for i in range(0, 300, 11):
        sh.remove(i * 7); plain.remove(i * 7); rows.pop(i * 7)
    sh.remove(123456789)                                               # absent: Ok(())
    assert sh.len() == plain.len() == len(rows)
    some = sorted(rows)[17]
    assert np.array_equal(sh.get_vector(some).data, rows[some]) and sh.get_vector(999999) is None
    bulk = rng.random((50, d), dtype=np.float32)
    bulk_ids = np.array([sorted(rows)[j] for j in range(0, 50)], dtype=np.uint64)     # a bulk that OVERWRITES stored ids: their old rows go
    sh.add_bulk(bulk, ids=bulk_ids); plain.add_bulk(bulk, ids=bulk_ids)
    for j, i in enumerate(bulk_ids):
        rows[int(i)] = bulk[j]
    dup_ids = np.array([5000, 5001, 5000, 5002, 5001], dtype=np.uint64)               # the same id twice in one batch: last wins
    dup = rng.random((5, d), dtype=np.float32)
    sh.add_bulk(dup, ids=dup_ids); plain.add_bulk(dup, ids=dup_ids)
    rows[5000], rows[5001], rows[5002] = dup[2], dup[4], dup[3]
    assert sh.len() == plain.len() == len(rows)
    ids = np.array(sorted(rows), dtype=np.uint64)
    mat = np.stack([rows[int(i)] for i in ids])
    queries = rng.random((9, d), dtype=np.float32)
    ks = np.array([1, 3, 400, 2, 7, 5, 5, 1, 10], dtype=np.uintp)      # per-query k (storage.rs:304), one of them above len
    got = sh.search_batch_arrays(queries, ks)
    ref = plain.search_batch_arrays(queries, ks)
    assert _same(ref, got)
    for b in range(9):
        oi, od = oracle.flat_search(0, mat, queries[b], int(ks[b]), ids=ids)
        assert got[2][b] == len(oi) and np.array_equal(got[0][b, :len(oi)], oi) and np.array_equal(got[1][b, :len(od)].view(np.uint32), od.view(np.uint32))


def test_sharded_handle_fewer_rows_than_shards_and_empty(vdb):
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.DotProduct, devices=[0, 0, 0, 0])
    q = np.ones((2, 3), dtype=np.float32)
    ids, dists, counts = sh.search_batch_arrays(q, 5)                  # storage.rs:218-220: empty store -> Ok(vec![])
    assert list(counts) == [0, 0]
    sh.add(4, vdb.Vector([1.0, 0.0, 0.0])); sh.add(2, vdb.Vector([0.0, 1.0, 0.0])); sh.add(9, vdb.Vector([1.0, 1.0, 0.0]))
    assert sorted(sh.shard_len(g) for g in range(4)) == [0, 1, 1, 1]
    res = sh.search(vdb.Vector([1.0, 0.0, 0.0]), 5)                    # k > len: len results
    assert [r[0] for r in res] == [4, 9, 2] and [float(r[1]) for r in res] == [-1.0, -1.0, 0.0]      # -dot, ties by id


def test_sharded_handle_errors_keep_the_reference_semantics(vdb):
    rng = np.random.default_rng(8)
    n, d = 2000, 16
    rows = rng.random((n, d), dtype=np.float32)
    rows[n - 1] = 0.0                                                  # ONE zero-norm row, on the last shard only
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Cosine, devices=[0, 0])
    sh.add_bulk(rows)
    q = rng.random((3, d), dtype=np.float32)
    with pytest.raises(vdb.InvalidVector):                             # distance.rs:51-55 aborts the whole search (flat_index.rs:57-60)
        sh.search_batch_arrays(q, 4)
    sh.remove(n - 1)
    got = sh.search_batch_arrays(q, 4)
    _check_oracle(1, rows[:n - 1], q, 4, got)
    with pytest.raises(vdb.DimensionMismatch) as e:                    # distance.rs:21-26: expected = query dim, actual = row dim
        sh.search_batch_arrays(np.ones((1, d + 1), dtype=np.float32), 4)
    assert (e.value.expected, e.value.actual) == (d + 1, d)
    qz = q.copy(); qz[1] = 0.0
    with pytest.raises(vdb.InvalidVector):                             # zero-norm query
        sh.search_batch_arrays(qz, 4)
    got2 = sh.search_batch_arrays(q, 4)                                # the handle stays usable after every error
    assert _same(got, got2)


def test_sharded_handle_id_mask_prefilter(vdb):
    rng = np.random.default_rng(9)
    n, d, k = 70_000, 32, 10
    rows = rng.random((n, d), dtype=np.float32)
    queries = rng.random((5, d), dtype=np.float32)
    live = (np.arange(n) % 4 == 1).astype(np.uint8)
    mask = np.packbits(live, bitorder="little")
    mask = np.concatenate([mask, np.zeros((-len(mask)) % 8, dtype=np.uint8)]).view(np.uint64)
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 0], keep_host_copy=False)
    sh.add_bulk(rows)
    got = sh.search_batch_arrays(queries, k, id_mask=mask, mask_bits=n)
    _check_oracle(0, rows, queries, k, got, live=live)


def test_sharded_handle_refuses_what_it_does_not_offer(vdb):
    L = vdb._ffi.lib()
    sh = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 0])
    sh.add(1, vdb.Vector([1.0, 2.0]))
    t = ctypes.c_int(-1)
    z = ctypes.c_void_p(8)
    assert L.vdb_flat_search_batch_device_submit(sh._h, z, 1, 2, 1, None, 0, z, z, z, None, ctypes.byref(t)) == vdb._ffi.ERR_INVALID_ARGUMENT
    assert "sharded handle" in vdb._ffi.last_error()[0]
    assert L.vdb_flat_search_batch_device_begin(sh._h, z, 1, 2, 1, None, 0, z, z, z, None, None) == vdb._ffi.ERR_INVALID_ARGUMENT
    with pytest.raises(vdb.IndexError_):
        sh.set_exchange(sh.EXCHANGE_RCCL)                              # a device listed twice: RCCL cannot serve it
    with pytest.raises(vdb.IndexError_):
        vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 99])
    with pytest.raises(vdb.IndexError_):
        vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[])
    plain = vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean)
    with pytest.raises(vdb.IndexError_):
        plain.set_exchange(0)


def test_store_over_a_sharded_index_replays_the_reference_fixtures(vdb, known_answers):
    """VectorStore::with_index(any Index) (storage.rs:118): the store tests of the reference over the sharded handle."""
    store = vdb.VectorStore.with_index(vdb.GpuFlatIndex(vdb.DistanceMetric.Euclidean, devices=[0, 0]))
    store.insert("v1", vdb.Vector([1.0, 0.0, 0.0]))
    store.insert("v2", vdb.Vector([0.0, 1.0, 0.0]))
    store.insert("v3", vdb.Vector([1.0, 1.0, 0.0]))
    res = store.search(vdb.Vector([1.0, 0.0, 0.0]), 2)                 # storage.rs:384-396
    assert res[0].id == "v1" and abs(res[0].distance) < 1e-6 and len(res) == 2
    out = store.search_batch([(vdb.Vector([1.0, 0.0, 0.0]), 1), (vdb.Vector([0.0, 1.0, 0.0]), 1)])   # storage.rs:680-697
    assert [[r.id for r in rs] for rs in out] == [["v1"], ["v2"]]
    store.delete("v1")
    assert store.len() == 2 and store.search(vdb.Vector([1.0, 0.0, 0.0]), 1)[0].id == "v3"
