"""BASELINE config 3 at FULL size on ONE GPU: 10M x 768 f32 (30.7 GB of the 288 GB), dot product, batch 1024, k = 100, as ONE
vdb_flat_create_sharded handle with 8 row shards on device 0 (peer exchange) -- every code path of the 8-GPU job except the
wires: routing of the rows, eight local searches (two 512-query passes each), the packed exchange, the merge.  Two queries are
compared with the oracle over all 10M rows (ids, order, distance bits); the rest against a plain single index over the same rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, oracle

def main():
    vdb = bench.load_package()
    dev = torch.device("cuda:0")
    n, d, B, k, G = 10_000_000, 768, 1024, 100, 8
    t0 = time.perf_counter()
    multi = vdb.GpuFlatIndex(vdb.DistanceMetric(2), devices=[0] * G, keep_host_copy=False)
    multi.reserve(n, d)
    per = n // G
    # the handle deals a bulk to its shards in contiguous blocks: one bulk of the whole matrix would need it resident twice, so
    # the rows go in shard by shard (block g of the ids lands on shard g through the routed add of an id range)
    for c in range((n + bench.CHUNK - 1) // bench.CHUNK):
        c0, c1 = c * bench.CHUNK, min((c + 1) * bench.CHUNK, n)
        block = bench.gen_chunk(c, c1 - c0, d, dev)
        torch.cuda.synchronize()
        multi.add_bulk_device(block.data_ptr(), c1 - c0, d, first_id=c0)
        del block
    multi.flush()
    print("built %d rows in %d shards %s in %.1f s" % (len(multi), multi.shards(), [multi.shard_len(g) for g in range(G)], time.perf_counter() - t0), flush=True)
    q = bench.gen_queries(B, d, dev)
    ids = torch.empty((B, k), dtype=torch.int64, device=dev); ds = torch.empty((B, k), dtype=torch.float32, device=dev); cn = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(2):
        multi.search_batch_device(q.data_ptr(), B, d, k, ids.data_ptr(), ds.data_ptr(), cn.data_ptr())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5):
        multi.search_batch_device(q.data_ptr(), B, d, k, ids.data_ptr(), ds.data_ptr(), cn.data_ptr())
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / 5
    st = multi.last_stats(); ss = multi.shard_stats()
    print("batch of %d, k = %d over %d shards on one GPU: %.2f ms (%.2f ms per shard); exchanges %d, uncertified %d, rows scanned %d" % (
        B, k, G, ms, ms / G, ss["exchanges"], st["uncertified"], st["rows_scanned"]), flush=True)
    assert int(cn.min()) == k
    gi = ids.cpu().numpy().astype(np.uint64); gd = ds.cpu().numpy()
    # oracle over all 10M rows for two queries
    rows = np.empty((n, d), dtype=np.float32)
    for c in range((n + bench.CHUNK - 1) // bench.CHUNK):
        c0, c1 = c * bench.CHUNK, min((c + 1) * bench.CHUNK, n)
        rows[c0:c1] = bench.gen_chunk(c, c1 - c0, d, dev).cpu().numpy()
    qh = q.cpu().numpy()
    for b in (0, B - 1):
        t2 = time.perf_counter()
        oi, od = oracle.flat_search(2, rows, qh[b], k)
        ok = np.array_equal(gi[b], oi) and np.array_equal(gd[b].view(np.uint32), od.view(np.uint32))
        print("query %d against the oracle over %d rows (%.1f s): %s" % (b, n, time.perf_counter() - t2, "ids, order and distance bits identical" if ok else "MISMATCH"), flush=True)
        assert ok
    print("C3 FULL SIZE OK")

if __name__ == "__main__":
    main()
