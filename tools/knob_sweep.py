"""One setting of the diagnostics build's tuning knobs per process (they are read when the index is created): the C2 step as
the median of back-to-back 20-step rounds after settling rounds (tools/clock_series.py: +-0.3 % repeatable).
Usage: VDB_LIB=.../libvdbflat_diag.so VDB_KP_FIRST=.. python tools/knob_sweep.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

def main():
    vdb = bench.load_package()
    dev = torch.device("cuda:0")
    index = bench.build_index(vdb, 1, 0, 1_000_000, 1_000_000, 768, dev, 0)
    q = bench.gen_queries(256, 768, dev)
    from vectordb_from_scratch_amd.sharded import gpu_local_search
    search = gpu_local_search(index, reuse_outputs=True)
    step = lambda: search(q, 10)
    def rnd(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
    for _ in range(4): rnd(20)
    r = [rnd(20) for _ in range(10)]
    st = index.last_stats()
    print("%-28s median %.4f  min %.4f  max %.4f ms/step   uncertified %d f32 %d overflows %d kprime %s" % (
        sys.argv[1] if len(sys.argv) > 1 else "default", float(np.median(r)), min(r), max(r), st.get("uncertified", -1),
        st.get("f32_tier_queries", -1), st.get("pool_overflows", -1), st.get("kprime")), flush=True)

if __name__ == "__main__":
    main()
