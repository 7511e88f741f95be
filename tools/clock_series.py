"""Time series of 20-step rounds of the C2 step in ONE process: what a `--steps 20 --warmup 5` measurement sees depending on when
it is taken.  Prints (seconds since the first search, kind, ms per step) for: a few f32-tier steps (bench.py's first leg), then
rounds of the synchronous step back to back, then the same with an idle pause in front of each round, then pipelined rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    import importlib
    vdb = bench.load_package()
    dev = torch.device("cuda:0")
    index = bench.build_index(vdb, 1, 0, 1_000_000, 1_000_000, 768, dev, 0)
    q = bench.gen_queries(256, 768, dev)
    from vectordb_from_scratch_amd.sharded import gpu_local_search
    search = gpu_local_search(index, reuse_outputs=True)
    step = lambda: search(q, 10)
    T0 = time.perf_counter()
    def rnd(kind, warm, n, fn=step):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        e = time.perf_counter()
        print("%8.3f s  %-28s %.4f ms/step" % (t - T0, kind, 1e3 * (e - t) / n), flush=True)
    index.set_screen(0); rnd("f32 tier", 2, 10); index.set_screen(1)
    for i in range(12): rnd("sync back-to-back", 5, 20)
    for pause in (0.005, 0.02, 0.1, 0.5):
        for i in range(3):
            time.sleep(pause); rnd("sync after %.3f s idle" % pause, 5, 20)
    for i in range(4): rnd("sync 100 steps", 5, 100)
    for i in range(6): rnd("sync back-to-back", 5, 20)

if __name__ == "__main__":
    main()
