"""BASELINE's full size (1M x 768 f32, batch 256, k = 10) on the GPU, checked through size-independent
properties plus a few whole queries against the CPU oracle (each costs seconds of CPU)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_package

pytestmark = pytest.mark.gpu

N, D, B, K = 1_000_000, 768, 256, 10


@pytest.fixture(scope="module")
def big():
    vdb = load_package()
    vdb.build()
    from vectordb_from_scratch_amd.sharded import gpu_local_search, merge_topk_hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rows = torch.rand((N, D), generator=g, device=dev, dtype=torch.float32)
    queries = torch.rand((B, D), generator=g, device=dev, dtype=torch.float32)
    queries[:8] = rows[[0, 1, 999_999, 500_000, 31, 32, 127, 128]]          # self queries
    return dict(vdb=vdb, rows=rows, queries=queries, local=gpu_local_search, merge=merge_topk_hip)


def build(vdb, metric, rows, first_id=0, screen=1):
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    ix.set_screen(screen)
    ix.add_bulk_device(rows.data_ptr(), rows.shape[0], rows.shape[1], first_id=first_id)
    ix.flush()
    return ix


@pytest.mark.parametrize("screen", [1, 0], ids=["bf16-screen", "f32-tier"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_full_size_properties(big, metric, screen):
    vdb, rows, queries = big["vdb"], big["rows"], big["queries"]
    ix = build(vdb, metric, rows, screen=screen)
    ids, dists, counts = big["local"](ix)(queries, K)
    torch.cuda.synchronize()
    st = ix.last_stats()
    assert st["bf16_screen"] == screen, st
    assert st["rows_scanned"] >= N and st["pool_overflows"] == 0, st
    assert st["exact_queries"] == 0, st                                     # every query certified on an MFMA tier
    if screen:
        assert st["f32_tier_queries"] <= 8, st                               # (nearly) every query certified by the screening tier
    assert torch.all(counts == K)
    assert torch.all(dists[:, 1:] >= dists[:, :-1])                          # ascending
    assert torch.all((ids >= 0) & (ids < N))
    assert all(len(set(r.tolist())) == K for r in ids.cpu())                 # no duplicate ids
    self_rows = [0, 1, 999_999, 500_000, 31, 32, 127, 128]
    if metric in (0, 1):                                                     # a stored row is its own nearest neighbour
        assert ids[:8, 0].tolist() == self_rows
    if metric == 0:
        assert torch.all(dists[:8, 0] == 0.0)                                # exact zero, not -6e-5
    # idempotence
    ids2, dists2, _ = big["local"](ix)(queries, K)
    assert torch.equal(ids, ids2) and torch.equal(dists, dists2)
    # two half-size shards merged == the whole index (linearity of the top-k merge)
    half = N // 2
    a = build(vdb, metric, rows[:half], screen=screen)
    b = build(vdb, metric, rows[half:], first_id=half, screen=screen)
    pa, pb = big["local"](a)(queries, K), big["local"](b)(queries, K)
    mi, md, mc = big["merge"](torch.stack([pa[0], pb[0]]), torch.stack([pa[1], pb[1]]),
                              torch.stack([pa[2], pb[2]]), K)
    torch.cuda.synchronize()
    assert torch.equal(mi, ids) and torch.equal(md, dists)
    # a few whole queries against the oracle: ids and distances bit-identical
    rows_h = rows.cpu().numpy()
    q_h = queries.cpu().numpy()
    for bq in (0, 8, 255):
        oi, od = oracle.flat_search(metric, rows_h, q_h[bq], K)
        assert np.array_equal(ids[bq].cpu().numpy().astype(np.uint64), oi)
        assert np.array_equal(dists[bq].cpu().numpy(), od)


def test_full_size_prefilter_and_k30(big):
    """Config-4 style: 25 % eq-filter as a device bitmask, and the 3k over-fetch k the reference's
    post-filter would issue (storage.rs:269)."""
    vdb, rows, queries = big["vdb"], big["rows"], big["queries"]
    ix = build(vdb, 0, rows)
    mask = np.zeros((N + 63) // 64, dtype=np.uint64)
    keep = np.arange(N) % 4 == 0
    np.bitwise_or.at(mask, np.nonzero(keep)[0] >> 6, np.uint64(1) << (np.nonzero(keep)[0] & 63).astype(np.uint64))
    q_h = queries[:16].cpu().numpy()
    gi, gd, gc = ix.search_batch_arrays(q_h, K, id_mask=mask, mask_bits=N)
    assert np.all(gc == K) and np.all(gi % 4 == 0)
    fi, fd, fc = ix.search_batch_arrays(q_h, 3 * K)                          # unfiltered, k = 30 -> kp = 64
    assert np.all(fc == 3 * K)
    rows_h = rows.cpu().numpy()
    for b in (0, 9):
        oi, od = oracle.flat_search(0, rows_h, q_h[b], K, live=keep.astype(np.uint8))
        assert np.array_equal(gi[b], oi) and np.array_equal(gd[b], od)
        # reference post-filter result (prefix property)
        post = [i for i in fi[b] if i % 4 == 0][:K]
        assert list(gi[b][:len(post)]) == post


@pytest.mark.parametrize("screen", [1, 0], ids=["bf16-screen", "f32-tier"])
def test_full_size_k100_stays_on_the_mfma_path(big, screen):
    """Config-3 style k = 100 (f32 tier: kp = 128; screening tier: up to 512 candidates): the sample must be sized
    so that no pool overflows and every query is certified by an MFMA tier (a too-small sample once sent 1019 of
    1024 queries to the exact fallback)."""
    vdb, rows, queries = big["vdb"], big["rows"], big["queries"]
    ix = build(vdb, 2, rows, screen=screen)
    q_h = queries[:64].cpu().numpy()
    gi, gd, gc = ix.search_batch_arrays(q_h, 100)
    st = ix.last_stats()
    assert st["kprime"] == (512 if screen else 128) and st["exact_queries"] == 0 and st["pool_overflows"] == 0, st
    assert np.all(gc == 100)
    rows_h = rows.cpu().numpy()
    for b in (3, 40):
        oi, od = oracle.flat_search(2, rows_h, q_h[b], 100)
        assert np.array_equal(gi[b], oi) and np.array_equal(gd[b], od)
