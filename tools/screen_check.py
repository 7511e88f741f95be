#!/usr/bin/env python3
"""A/B of the two tiers on one GPU: the bf16 screening tier (default) against the f32 MFMA tier on the same index
and queries.  Results must be identical bit for bit; prints step / kernel times and the path counters.

    python tools/screen_check.py [--rows N] [--dim D] [--batch B] [--k K] [--metric M] [--dist uniform|gauss]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402


def run(vdb, n, dim, B, k, metric, dist, steps=5, oracle_q=0):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), device=0, keep_host_copy=False)
    ix.reserve(n, dim)
    chunk = 125_000
    host = [] if oracle_q else None
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        blk = torch.rand((m, dim), generator=g, device=dev) if dist == "uniform" else torch.randn((m, dim), generator=g, device=dev)
        if dist == "gauss":
            blk = blk / blk.norm(dim=1, keepdim=True)
        torch.cuda.synchronize()
        ix.add_bulk_device(blk.data_ptr(), m, dim, first_id=c0)
        if host is not None:
            host.append(blk.cpu().numpy())
        del blk
    ix.flush()
    q = torch.rand((B, dim), generator=g, device=dev) if dist == "uniform" else torch.randn((B, dim), generator=g, device=dev)
    out = {}
    for mode in (1, 0):
        ix.set_screen(mode)
        ids = torch.empty((B, k), dtype=torch.int64, device=dev)
        ds = torch.empty((B, k), dtype=torch.float32, device=dev)
        cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        def step():
            ix.search_batch_device(q.data_ptr(), B, dim, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr())
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        st = ix.last_stats()
        ix.set_profile(True)
        step()
        kms = ix.last_stats()["fused_kernel_ns"] / 1e6
        ix.set_profile(False)
        out[mode] = (ids.cpu().numpy().copy(), ds.cpu().numpy().copy(), cnt.cpu().numpy().copy())
        print(f"  mode={'bf16' if mode else 'f32 '} step {ms:8.3f} ms  kernel {kms:8.3f} ms  "
              f"{4.0 * n * dim / (kms * 1e-3) / 1e12 if kms else 0:5.2f} TB/s(alg)  stats {st}", flush=True)
    same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(out[1], out[0]))
    print(f"  identical(ids, dist bits, counts): {same}", flush=True)
    ok = same
    if oracle_q:
        import oracle
        rows = np.concatenate(host)
        qh = q.cpu().numpy()
        for b in range(oracle_q):
            oi, od = oracle.flat_search(metric, rows, qh[b], k)
            e = np.array_equal(oi, out[1][0][b, :len(oi)].astype(np.uint64)) and np.array_equal(od.view(np.uint32), out[1][1][b, :len(od)].view(np.uint32))
            ok &= bool(e)
        print(f"  oracle parity on {oracle_q} queries: {ok}", flush=True)
    ix.close() if hasattr(ix, "close") else None
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--oracle", type=int, default=0)
    a = ap.parse_args()
    vdb = load_package()
    vdb.build()
    ok = True
    if a.rows:
        print(f"n={a.rows} dim={a.dim} B={a.batch} k={a.k} metric={a.metric} {a.dist}", flush=True)
        ok &= run(vdb, a.rows, a.dim, a.batch, a.k, a.metric, a.dist, oracle_q=a.oracle)
    else:
        for (n, dim, B, k, metric, dist) in [
            (20000, 64, 7, 3, 0, "uniform"), (50000, 32, 256, 10, 1, "gauss"), (100001, 100, 33, 10, 2, "uniform"),
            (200000, 768, 256, 10, 1, "uniform"), (200000, 768, 256, 10, 0, "uniform"), (200000, 768, 256, 10, 2, "gauss"),
            (300000, 1536, 300, 16, 0, "gauss"), (1000000, 768, 256, 10, 1, "uniform"), (1000000, 768, 256, 10, 0, "uniform"),
        ]:
            print(f"n={n} dim={dim} B={B} k={k} metric={metric} {dist}", flush=True)
            ok &= run(vdb, n, dim, B, k, metric, dist, oracle_q=2 if n <= 200000 else 0)
    print("ALL OK" if ok else "MISMATCH", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
