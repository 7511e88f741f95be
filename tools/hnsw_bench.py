#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: HnswIndex (m = 16, ef_construction = 200) with GPU-offloaded distance evaluation.
Build time, batched search throughput at ef_search, recall@10 against the exact GPU FlatIndex, and -- with
--oracle -- the same graph and queries on the CPU restatement (identical results, single-thread QPS).

    python tools/hnsw_bench.py [--rows N] [--dim D] [--batch B] [--ef EF] [--oracle]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--scan-build", action="store_true", help="the row-scan build of round 2 (vdb_hnsw_set_build(h, 0)) instead of the frontier-only build")
    a = ap.parse_args()
    vdb = load_package()
    vdb.build()
    rng = np.random.default_rng(1)
    rows = rng.random((a.rows, a.dim), dtype=np.float32)                 # the reference benches' distribution
    queries = rng.random((a.batch, a.dim), dtype=np.float32)
    ids = np.arange(a.rows, dtype=np.uint64)
    g = vdb.GpuHnswIndex(vdb.DistanceMetric(a.metric), vdb.HnswParams.new(a.m, a.efc, 50), seed=1)
    if a.scan_build:
        g.set_build(False)


    t0 = time.perf_counter()
    step = 20000
    for c0 in range(0, a.rows, step):
        g.build_batch((ids[c0:c0 + step], rows[c0:c0 + step]))
        print(f"  built {min(c0 + step, a.rows)} rows, {time.perf_counter() - t0:.1f} s", flush=True)
    t_build = time.perf_counter() - t0
    st = g.stats()
    print(f"build: {a.rows} x {a.dim}, m={a.m} ef_construction={a.efc}: {t_build:.1f} s "
          f"({1e3 * t_build / a.rows:.3f} ms per insert), GPU distances {st['gpu_distances']:.3e}, launches {st['gpu_launches']}", flush=True)
    print("build stats:", g.build_stats(), flush=True)
    print("build times:", g.build_times(), flush=True)
    g.search_batch_arrays(queries, a.k, a.ef)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        hi, hd, hc = g.search_batch_arrays(queries, a.k, a.ef)
    t_s = (time.perf_counter() - t0) / reps
    st = g.stats()
    flat = vdb.GpuFlatIndex(vdb.DistanceMetric(a.metric), keep_host_copy=False)
    flat.add_bulk(rows)
    ti, _, _ = flat.search_batch_arrays(queries, a.k)
    t0 = time.perf_counter()
    for _ in range(reps):
        flat.search_batch_arrays(queries, a.k)
    t_f = (time.perf_counter() - t0) / reps
    rec = np.mean([len(set(ti[b]) & set(hi[b, :hc[b]])) / float(a.k) for b in range(a.batch)])
    print(f"search: batch {a.batch}, k={a.k}, ef={a.ef}: {1e3 * t_s:.2f} ms per batch = {a.batch / t_s:.0f} queries/s; "
          f"device-resident walks {st['device_queries']}, host re-runs {st['host_redone']}, host traversal rounds of the last batch {st['last_search_rounds']}; "
          f"recall@{a.k} vs exact = {rec:.4f}; exact GPU FlatIndex on the same batch (host pointers): {a.batch / t_f:.0f} queries/s", flush=True)
    if a.oracle:
        import oracle
        o = oracle.HnswOracle(a.metric, m=a.m, ef_construction=a.efc, ef_search=50, seed=1)
        t0 = time.perf_counter()
        for i in range(a.rows):
            o.insert(i, rows[i])
        t_ob = time.perf_counter() - t0
        nq = min(a.batch, 64)
        t0 = time.perf_counter()
        same = True
        for b in range(nq):
            oi, od = o.search(queries[b], a.k, a.ef)
            same &= bool(np.array_equal(oi, hi[b, :hc[b]]) and np.array_equal(od.view(np.uint32), hd[b, :hc[b]].view(np.uint32)))
        t_os = time.perf_counter() - t0
        print(f"cpu restatement (1 core): build {t_ob:.1f} s, search {nq / t_os:.0f} queries/s; results identical to the GPU-offloaded index: {same}", flush=True)


if __name__ == "__main__":
    main()
