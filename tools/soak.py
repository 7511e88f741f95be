"""Soak: thousands of searches through the three call forms on one GPU -- synchronous, two batches in flight (submit / wait),
and ONE handle over four shards (peer exchange) -- with the results compared against the first answer at intervals and the
free device memory watched (no growth = no leak of workspaces, events or staging buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

def main():
    vdb = bench.load_package()
    dev = torch.device("cuda:0")
    n, d, B, k = 400_000, 768, 256, 10
    index = bench.build_index(vdb, 1, 0, n, n, d, dev, 0)
    q = bench.gen_queries(B, d, dev)
    from vectordb_from_scratch_amd.sharded import gpu_local_search
    search = gpu_local_search(index, reuse_outputs=True)
    ref = tuple(t.clone() for t in search(q, k))
    def same(o):
        return bool(torch.equal(o[0], ref[0]) and torch.equal(o[1].view(torch.int32), ref[1].view(torch.int32)) and torch.equal(o[2], ref[2]))
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    for i in range(3000):
        o = search(q, k)
        if i % 500 == 499: assert same(o), ("sync", i)
    torch.cuda.synchronize()
    print("sync      3000 steps %.2f s, free memory change %d MiB" % (time.perf_counter() - t0, (torch.cuda.mem_get_info()[0] - free0) >> 20), flush=True)
    bufs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(2)]
    def submit(i):
        b = bufs[i & 1]
        return index.search_batch_device_submit(q.data_ptr(), B, d, k, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
    t0 = time.perf_counter()
    t = submit(0)
    for i in range(1, 3000):
        t2 = submit(i)
        index.search_batch_device_wait(t)
        if i % 500 == 0: assert same(bufs[(i - 1) & 1]), ("pipelined", i)
        t = t2
    index.search_batch_device_wait(t)
    print("pipelined 3000 steps %.2f s, free memory change %d MiB" % (time.perf_counter() - t0, (torch.cuda.mem_get_info()[0] - free0) >> 20), flush=True)
    # mutations between searches: remove and re-add a row; the answers before and after must agree with the first
    rows = bench.gen_chunk(0, 4, d, dev).cpu().numpy()
    for i in range(200):
        index.remove(5); index.add(5, vdb.Vector(rows[min(5, 3)] if False else bench.gen_chunk(0, 8, d, dev)[5].cpu().numpy()))
        o = search(q, k)
    assert same(o), "after mutations"
    print("mutations 200 remove/add + search ok, free memory change %d MiB" % ((torch.cuda.mem_get_info()[0] - free0) >> 20), flush=True)
    del index, search
    multi = vdb.GpuFlatIndex(vdb.DistanceMetric(1), devices=[0, 0, 0, 0], keep_host_copy=False)
    multi.reserve(n, d)
    chunk = min(bench.CHUNK, n)
    for c in range((n + chunk - 1) // chunk):
        c0, c1 = c * chunk, min((c + 1) * chunk, n)
        block = bench.gen_chunk(c, c1 - c0, d, dev)
        torch.cuda.synchronize()
        multi.add_bulk_device(block.data_ptr(), c1 - c0, d, first_id=c0)
    multi.flush()
    msearch = gpu_local_search(multi, reuse_outputs=True)
    free1 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    for i in range(1000):
        o = msearch(q, k)
        if i % 250 == 249: assert same(o), ("sharded", i)
    torch.cuda.synchronize()
    print("4 shards  1000 steps %.2f s (%s), free memory change %d MiB" % (time.perf_counter() - t0, multi.shard_stats() if hasattr(multi, "shard_stats") else "", (torch.cuda.mem_get_info()[0] - free1) >> 20), flush=True)
    print("SOAK OK")

if __name__ == "__main__":
    main()
