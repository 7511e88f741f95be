#!/usr/bin/env python3
"""Randomised differential test of the HNSW index: random sizes, dimensions, metrics, m / ef_construction / ef, data with
duplicates, interleaved removals and re-inserts; the product's graph must equal the CPU restatement's (same seed) and
every search result must be identical (ids, order, distance bits) -- through the device-resident search.

    python tools/fuzz_hnsw.py [--cases N] [--seed S]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    vdb = load_package()
    vdb.build()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    dev_q = host_q = 0
    for case in range(a.cases):
        n = int(rng.integers(50, 4000))
        d = int(rng.choice([1, 2, 7, 16, 33, 64, 128, 300]))
        metric = int(rng.integers(0, 3))
        m = int(rng.choice([2, 4, 8, 16, 19, 24]))
        efc = int(rng.choice([8, 32, 100, 200]))
        kind = str(rng.choice(["uniform", "gauss", "dups"]))
        if kind == "uniform":
            rows = rng.random((n, d), dtype=np.float32)
        elif kind == "gauss":
            rows = rng.standard_normal((n, d)).astype(np.float32)
        else:
            base = rng.random((max(n // 8, 1), d), dtype=np.float32)
            rows = base[rng.integers(0, base.shape[0], n)]
        if metric == 1:
            rows[np.linalg.norm(rows, axis=1) == 0] += 1.0
        seed = int(rng.integers(1, 1 << 30))
        g = vdb.GpuHnswIndex(vdb.DistanceMetric(metric), vdb.HnswParams.new(m, efc, 50), seed=seed)
        o = oracle.HnswOracle(metric, m=m, ef_construction=efc, ef_search=50, seed=seed)
        ids = rng.permutation(n * 2)[:n]
        half = n // 2
        g.build_batch((ids[:half].astype(np.uint64), rows[:half]))
        for i in range(half):
            o.insert(int(ids[i]), rows[i])
        removed = set()
        for v in rng.choice(ids[:half], size=min(half, int(rng.integers(0, 40))), replace=False):
            g.remove(int(v)); o.remove(int(v)); removed.add(int(v))
        for i in range(half, n):                                   # single adds after the removals
            g.add(int(ids[i]), vdb.Vector(rows[i]))
            o.insert(int(ids[i]), rows[i])
        desc = f"case {case}: n={n} d={d} metric={metric} m={m} efc={efc} data={kind} removed={len(removed)}"
        ok = g.len() == len(o) and g.entry_point() == o.entry_point()
        for i in ids:
            lv = o.level(int(i))
            ok &= g.level(int(i)) == lv
            for l in range(max(lv, -1) + 1):
                ok &= g.neighbors(int(i), l) == o.neighbors(int(i), l)
        nq = int(rng.choice([1, 5, 64]))
        q = rng.standard_normal((nq, d)).astype(np.float32) if kind == "gauss" else rng.random((nq, d), dtype=np.float32)
        if metric == 1:
            q[np.linalg.norm(q, axis=1) == 0] += 1.0
        for (k, ef) in [(10, 100), (1, 16), (int(rng.integers(1, 60)), int(rng.choice([10, 50, 300])))]:
            gi, gd, gc = g.search_batch_arrays(q, k, ef)
            for b in range(nq):
                oi, od = o.search(q[b], k, ef)
                ok &= bool(gc[b] == len(oi) and np.array_equal(gi[b, :gc[b]], oi) and np.array_equal(gd[b, :gc[b]].view(np.uint32), od.view(np.uint32)))
        st = g.stats()
        dev_q += st["device_queries"]; host_q += st["host_redone"]
        print(("ok   " if ok else "FAIL ") + desc + f"  [device walks {st['device_queries']} host re-runs {st['host_redone']} host rounds {st['last_search_rounds']}]", flush=True)
        if not ok:
            sys.exit(1)
    print(f"ALL {a.cases} HNSW CASES OK in {time.time() - t0:.0f} s; device-resident walks {dev_q}, host re-runs {host_q}", flush=True)


if __name__ == "__main__":
    main()
