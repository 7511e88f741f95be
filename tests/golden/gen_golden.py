#!/usr/bin/env python3
"""Generate tests/golden/flat_cases.npz: seeded inputs + the CPU oracle's outputs.

The reference (Rust) cannot be built in this image, so these
vectors are produced by the oracle restatement (oracle/flat_oracle.c), which is
itself pinned by reference_known_answers.json, and every distance is
cross-checked here against an independent float64 numpy computation before it
is written.  Run from the repo root:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

METRICS = {"euclidean": oracle.EUCLIDEAN, "cosine": oracle.COSINE, "dot": oracle.DOT}


def f64_dist(metric, rows, q):
    r = rows.astype(np.float64)
    v = q.astype(np.float64)
    if metric == "euclidean":
        return np.sqrt(((r - v) ** 2).sum(1))
    if metric == "dot":
        return -(r @ v)
    n = np.linalg.norm(r, axis=1) * np.linalg.norm(v)
    return 1.0 - np.clip((r @ v) / n, -1.0, 1.0)


def make_cases():
    cases = {}
    rng = np.random.default_rng(1234)
    # uniform[0,1): the reference benches' distribution (benches/search_bench.rs:6-13)
    cases["uniform"] = dict(rows=rng.random((4096, 64), dtype=np.float32),
                            queries=rng.random((16, 64), dtype=np.float32),
                            ids=np.arange(4096, dtype=np.uint64), ks=[1, 10, 100])
    # unit gaussian rows, odd dimension (exercises the K padding), sparse non-monotonic ids
    g = rng.standard_normal((1500, 50)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True).astype(np.float32)
    gq = rng.standard_normal((8, 50)).astype(np.float32)
    ids = rng.permutation(np.arange(10, 10 + 3 * 1500, 3)).astype(np.uint64)
    cases["gauss_sparse_ids"] = dict(rows=g, queries=gq, ids=ids, ks=[1, 10, 30])
    # near duplicates: pairs differ by ~1 ulp-level noise; queries are stored rows
    base = rng.random((300, 32), dtype=np.float32)
    dup = base + (rng.random((300, 32), dtype=np.float32) - 0.5) * 2e-7
    nd = np.concatenate([base, dup.astype(np.float32)], 0)
    cases["near_dup"] = dict(rows=nd, queries=nd[[0, 17, 299, 300, 555]].copy(),
                             ids=np.arange(600, dtype=np.uint64), ks=[1, 10])
    # integer grid: many exact ties, decided by lower id
    grid = rng.integers(0, 3, size=(512, 8)).astype(np.float32)
    cases["ties"] = dict(rows=grid, queries=rng.integers(0, 3, size=(6, 8)).astype(np.float32) + 0.0,
                         ids=rng.permutation(512).astype(np.uint64), ks=[1, 10, 40])
    # identical rows: everything ties
    same = np.tile(rng.random((1, 16), dtype=np.float32), (100, 1))
    cases["all_same"] = dict(rows=same, queries=rng.random((3, 16), dtype=np.float32),
                             ids=np.arange(100, dtype=np.uint64)[::-1].copy(), ks=[5, 100, 150])
    return cases


def main():
    out = {}
    for name, c in make_cases().items():
        rows, queries, ids = c["rows"], c["queries"], c["ids"]
        out[f"{name}/rows"] = rows
        out[f"{name}/queries"] = queries
        out[f"{name}/ids"] = ids
        out[f"{name}/ks"] = np.array(c["ks"], dtype=np.int64)
        for mname, m in METRICS.items():
            if mname == "cosine" and name == "ties":
                # grid data may contain zero rows -> InvalidVector; recorded as an error case
                zero = (np.abs(rows).sum(1) == 0).any() or (np.abs(queries).sum(1) == 0).any()
                out[f"{name}/{mname}/has_zero"] = np.array([int(zero)])
                if zero:
                    continue
            for k in c["ks"]:
                kk = min(k, rows.shape[0])
                eid = np.zeros((queries.shape[0], kk), dtype=np.uint64)
                ed = np.zeros((queries.shape[0], kk), dtype=np.float32)
                for b in range(queries.shape[0]):
                    i, d = oracle.flat_search(m, rows, queries[b], k, ids=ids)
                    assert i.size == kk
                    eid[b], ed[b] = i, d
                    # cross-check vs float64: distances agree, and the oracle's k-th distance
                    # is not beaten by any excluded row by more than f32 noise
                    ref = f64_dist(mname, rows, queries[b])
                    pos = {int(v): j for j, v in enumerate(ids)}
                    sel = np.array([pos[int(v)] for v in i])
                    scale = max(1.0, float(np.abs(ref).max()))
                    assert np.allclose(d, ref[sel], rtol=2e-5, atol=2e-6 * scale), (name, mname, k, b)
                    excl = np.ones(rows.shape[0], bool)
                    excl[sel] = False
                    if excl.any():
                        assert ref[excl].min() >= ref[sel].max() - 1e-5 * scale, (name, mname, k, b)
                out[f"{name}/{mname}/k{k}/ids"] = eid
                out[f"{name}/{mname}/k{k}/dists"] = ed
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flat_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
