#!/usr/bin/env python3
"""Randomised differential test on one GPU: random index shapes, metrics, data distributions, tombstones, id orders,
pre-filter masks, k and batch sizes through BOTH tier configurations of the engine, compared bit for bit with each other
and (on a sample of queries) with the CPU oracle.  Any mismatch prints the failing configuration and exits non-zero.

    python tools/fuzz_parity.py [--cases N] [--seed S] [--max-rows R] [--shadow]

--shadow additionally runs every case with the opt-in bf16 shadow rows (vdb_flat_set_shadow) and with the sample cache off
(vdb_flat_set_sample_cache(0)): same results, same tier counters.
--round3 additionally runs every case (i) with the 512-query filter kernel off (vdb_flat_set_wide(0): batches above 256
queries), (ii) with the direct path of small indexes off (VDB_TIERS_NO_DIRECT), and (iii) through ONE sharded handle over 2-4
shards on device 0 (vdb_flat_create_sharded, peer exchange): same results; it also draws small indexes (3 .. 16384 rows),
batches up to 1100 queries and per-query k.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402
import oracle  # noqa: E402


def make_data(rng, n, d, kind):
    if kind == "uniform":
        return rng.random((n, d), dtype=np.float32)
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "unit":
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    if kind == "clustered":                       # cluster by cluster in storage order, mixed scales
        c = rng.standard_normal((max(n // 500, 1), d)).astype(np.float32) * 3.0
        lab = np.sort(rng.integers(0, c.shape[0], n))
        return (c[lab] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    if kind == "dups":                            # many exact duplicates
        base = rng.random((max(n // 40, 1), d), dtype=np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "scales":                          # norms spread over two orders of magnitude
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x * np.exp(rng.uniform(-2.3, 2.3, (n, 1))).astype(np.float32)
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-rows", type=int, default=300000)
    ap.add_argument("--shadow", action="store_true")
    ap.add_argument("--round3", action="store_true")
    a = ap.parse_args()
    vdb = load_package()
    vdb.build()
    rng = np.random.default_rng(a.seed)
    kinds = ["uniform", "gauss", "unit", "clustered", "dups", "scales"]
    t0 = time.time()
    tiers = {"screen": 0, "rethr": 0, "f32q": 0, "exact": 0, "ovf": 0}
    for case in range(a.cases):
        n = int(rng.integers(16385, a.max_rows))
        if a.round3 and rng.random() < 0.3:
            n = int(rng.integers(3, 16385))                           # the dense-scores / direct paths
        d = int(rng.choice([1, 3, 17, 32, 33, 64, 100, 128, 200, 384, 768]))
        if n * d > 120_000_000:
            n = 120_000_000 // d
        metric = int(rng.integers(0, 3))
        kind = str(rng.choice(kinds))
        nq = int(rng.choice([1, 2, 31, 32, 33, 100, 256, 257, 300] + ([7, 8, 9, 511, 512, 513, 700, 1024, 1100] if a.round3 else [])))
        k = int(rng.choice([1, 2, 10, 10, 10, 30, 48, 49, 100, 112, 113, 150]))
        rows = make_data(rng, n, d, kind)
        if metric == 1:
            rows[np.linalg.norm(rows, axis=1) == 0] += 1.0      # a zero-norm row fails every Cosine search (tested elsewhere)
        queries = make_data(rng, nq, d, kind if kind not in ("clustered", "dups") else "gauss")
        if kind in ("clustered", "dups") and nq > 1:
            queries[: nq // 2] = rows[rng.integers(0, n, nq // 2)] + (1e-3 * rng.standard_normal((nq // 2, d))).astype(np.float32)
        if metric == 1:
            queries[np.linalg.norm(queries, axis=1) == 0] += 1.0
        ids = None
        if rng.random() < 0.3:
            ids = rng.permutation(n * 2)[:n].astype(np.uint64)  # sparse, non-monotone ids
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows, ids=ids)
        live = np.ones(n, dtype=np.uint8)
        if rng.random() < 0.4:                                   # tombstones
            dead = rng.choice(n, size=int(n * rng.uniform(0.001, 0.3)), replace=False)
            for r in dead[:2000]:
                ix.remove(int(ids[r]) if ids is not None else int(r))
                live[r] = 0
        mask = None
        elig = live.copy()
        if rng.random() < 0.3:                                   # pre-filter bitmask over ids
            sel = float(rng.choice([0.5, 0.25, 0.02, 0.001]))
            keep_rows = rng.random(n) < sel
            id_of = ids if ids is not None else np.arange(n, dtype=np.uint64)
            bits = int(id_of.max()) + 1
            mask = np.zeros((bits + 63) // 64, dtype=np.uint64)
            kid = id_of[keep_rows]
            np.bitwise_or.at(mask, (kid >> np.uint64(6)).astype(np.int64), np.uint64(1) << (kid & np.uint64(63)))
            elig = elig & keep_rows.astype(np.uint8)
        kw = dict(id_mask=mask, mask_bits=(int((ids if ids is not None else np.arange(n)).max()) + 1) if mask is not None else 0) if mask is not None else {}
        desc = f"case {case}: n={n} d={d} metric={metric} data={kind} nq={nq} k={k} ids={'perm' if ids is not None else 'seq'} dead={int((live == 0).sum())} mask={'yes' if mask is not None else 'no'}"
        ix.set_screen(1)
        a1 = ix.search_batch_arrays(queries, k, **kw)
        st = ix.last_stats()
        ix.set_screen(0)
        a0 = ix.search_batch_arrays(queries, k, **kw)
        ok = all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a1, a0))
        if a.shadow:
            ctr = ("bf16_screen", "uncertified", "rethreshold_queries", "f32_tier_queries", "exact_queries", "pool_overflows")
            ix.set_screen(1)
            ix.set_shadow(True)
            a2 = ix.search_batch_arrays(queries, k, **kw)
            s2 = ix.last_stats()
            ix.set_shadow(False)
            ix.set_sample_cache(False)
            a3 = ix.search_batch_arrays(queries, k, **kw)
            s3 = ix.last_stats()
            ix.set_sample_cache(True)
            for ax, sx in ((a2, s2), (a3, s3)):
                ok &= all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a1, ax))
                ok &= all(sx[c] == st[c] for c in ctr)
        if a.round3:
            same = lambda u, v: all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(u, v))   # noqa: E731
            ix.set_screen(1)
            ix.set_wide(False)
            ok &= same(a1, ix.search_batch_arrays(queries, k, **kw))
            ix.set_wide(True)
            ix.set_tiers(ix.TIERS_NO_DIRECT)
            ok &= same(a1, ix.search_batch_arrays(queries, k, **kw))
            ix.set_tiers(0)
            G = int(rng.integers(2, 5))
            sh = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), devices=[0] * G, keep_host_copy=False)
            half = n // 2
            sh.add_bulk(rows[:half], ids=None if ids is None else ids[:half], first_id=0)            # two bulks: the second on top of the first
            sh.add_bulk(rows[half:], ids=None if ids is None else ids[half:], first_id=half)
            for r in np.nonzero(live == 0)[0]:
                sh.remove(int(ids[r]) if ids is not None else int(r))
            ok &= same(a1, sh.search_batch_arrays(queries, k, **kw))
            if nq > 1 and mask is None:                                                           # per-query k: a prefix of the batch-wide result
                ks = rng.integers(0, k + 1, nq).astype(np.uintp)
                pi, pd, pc = sh.search_batch_arrays(queries, ks)
                for b in range(0, nq, max(1, nq // 7)):
                    kb = int(min(ks[b], a1[2][b]))
                    ok &= bool(pc[b] == kb and np.array_equal(pi[b, :kb], a1[0][b, :kb]) and np.array_equal(pd[b, :kb].view(np.uint32), a1[1][b, :kb].view(np.uint32)))
            del sh
        for b in sorted({0, nq // 2, nq - 1}):
            oi, od = oracle.flat_search(metric, rows, queries[b], k, ids=ids, live=elig)
            gi, gd, gc = a1
            ok &= bool(gc[b] == len(oi) and np.array_equal(gi[b, :gc[b]], oi) and np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)))
        tiers["screen"] += st["bf16_screen"]; tiers["rethr"] += st["rethreshold_queries"]; tiers["f32q"] += st["f32_tier_queries"]; tiers["exact"] += st["exact_queries"]; tiers["ovf"] += st["pool_overflows"]
        print(("ok   " if ok else "FAIL ") + desc + f"  [screen={st['bf16_screen']} rethr={st['rethreshold_queries']} f32q={st['f32_tier_queries']} exact={st['exact_queries']} ovf={st['pool_overflows']}]", flush=True)
        if not ok:
            sys.exit(1)
        del ix
    print(f"ALL {a.cases} CASES OK in {time.time() - t0:.0f} s; tier usage {tiers}", flush=True)


if __name__ == "__main__":
    main()
