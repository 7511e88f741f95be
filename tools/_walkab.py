import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import bench
vdb = bench.load_package()
rng = np.random.default_rng(1)
n, d = 60000, 768
x = rng.random((n, d), dtype=np.float32)
g = vdb.GpuHnswIndex(vdb.DistanceMetric(0), vdb.HnswParams.new(16, 200, 50), seed=7)
g.build_batch((np.arange(n, dtype=np.uint64), x))
q = rng.random((256, d), dtype=np.float32)
for i in range(4):
    t = time.perf_counter(); r = g.search_batch_arrays(q, 10, 200); print("batch ms", 1e3 * (time.perf_counter() - t), flush=True)
print("checksum", int(r[0].sum()) , float(r[1].sum()))
