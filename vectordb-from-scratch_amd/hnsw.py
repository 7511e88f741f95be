"""Host-side mirror of the reference's `HnswIndex` / `HnswParams` (src/hnsw/mod.rs:14-82, src/hnsw/graph.rs:19-59)
over the C ABI of include/vdb_hnsw.h: the graph lives on the host side of the library, every distance the
traversal asks for is evaluated on the MI355X (BASELINE config 5).  There is no CPU distance path."""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .index import Index, _fp, _raise, _u64p
from .vector import DistanceMetric, Vector


@dataclass
class HnswParams:
    """graph.rs:19-59: m_max0 = 2m, ml = 1/ln(m), max_layers = 16 are derived."""
    m: int = 16
    ef_construction: int = 200
    ef_search: int = 50

    @staticmethod
    def new(m, ef_construction, ef_search):          # HnswParams::new (graph.rs:49-59)
        return HnswParams(m, ef_construction, ef_search)


class GpuHnswIndex(Index):
    """Drop-in for HnswIndex (src/hnsw/mod.rs:14-82).  `seed` replaces the reference's StdRng::from_entropy()
    (graph.rs:101) for the node levels; with the same seed and insertion order the graph and all results equal the
    CPU restatement's (the test oracle)."""

    def __init__(self, metric, params=None, seed=1, device=0):
        self._metric = DistanceMetric(metric)
        self.params = params or HnswParams()
        self._L = _ffi.lib()
        self._h = ctypes.c_void_p()
        rc = self._L.vdb_hnsw_create(int(self._metric), self.params.m, self.params.ef_construction,
                                     self.params.ef_search, int(seed), int(device), ctypes.byref(self._h))
        if rc:
            _raise(rc)
        self._vectors = {}

    @classmethod
    def with_params(cls, metric, params, seed=1, device=0):      # mod.rs:28-32
        return cls(metric, params, seed=seed, device=device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.vdb_hnsw_destroy(h)

    # ---- Index trait (mod.rs:56-82)
    def add(self, id, vector, level=-1):
        v = vector if isinstance(vector, Vector) else Vector(vector)
        rc = self._L.vdb_hnsw_add(self._h, int(id), _fp(v.data), v.dimension(), int(level))
        if rc:
            _raise(rc)
        self._vectors[int(id)] = v

    def build_batch(self, vectors):                              # mod.rs:37-42
        """vectors: list of (id, Vector) -- or (ids array, rows [n, d] array) for the bulk form."""
        if isinstance(vectors, tuple) and len(vectors) == 2 and hasattr(vectors[1], "shape"):
            ids = np.ascontiguousarray(vectors[0], dtype=np.uint64)
            rows = np.ascontiguousarray(vectors[1], dtype=np.float32)
        else:
            ids = np.array([int(i) for i, _ in vectors], dtype=np.uint64)
            rows = np.ascontiguousarray(np.stack([(v.data if isinstance(v, Vector) else np.asarray(v, np.float32))
                                                  for _, v in vectors]), dtype=np.float32) if len(vectors) else np.zeros((0, 0), np.float32)
        if rows.shape[0] == 0:
            return
        rc = self._L.vdb_hnsw_add_bulk(self._h, _u64p(ids), 0, _fp(rows), rows.shape[0], rows.shape[1])
        if rc:
            _raise(rc)

    def remove(self, id):
        rc = self._L.vdb_hnsw_remove(self._h, int(id))
        if rc:
            _raise(rc)
        self._vectors.pop(int(id), None)

    def search(self, query, k):                                  # mod.rs:69-72: ef is fixed at 50
        return self.search_with_ef(query, k, 50)

    def search_with_ef(self, query, k, ef):                      # mod.rs:45-53
        q = query if isinstance(query, Vector) else Vector(query)
        ids, ds, cnt = self.search_batch_arrays(q.data.reshape(1, -1), k, ef)
        return [(int(ids[0, i]), float(ds[0, i])) for i in range(int(cnt[0]))]

    def search_batch_arrays(self, queries, k, ef=0):
        """queries [nq, dim] f32 -> (ids u64 [nq, k], dists f32 [nq, k], counts [nq]); every query of the batch walks
        the graph in lockstep, one GPU launch per traversal round for all their candidate lists."""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = qs.shape
        kk = max(int(k), 1)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        ds = np.zeros((nq, kk), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uintp)
        rc = self._L.vdb_hnsw_search_batch(self._h, _fp(qs), nq, dim, int(k), int(ef), _u64p(ids), _fp(ds),
                                           cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)))
        if rc:
            _raise(rc)
        return ids, ds, cnt

    def search_batch(self, queries):
        if not queries:
            return []
        qs = np.stack([(q.data if isinstance(q, Vector) else np.asarray(q, np.float32)) for q, _ in queries])
        kmax = max(k for _, k in queries)
        ids, ds, cnt = self.search_batch_arrays(qs, kmax, 50)
        return [[(int(ids[b, i]), float(ds[b, i])) for i in range(min(int(cnt[b]), k))] for b, (_, k) in enumerate(queries)]

    def get_vector(self, id):                                    # mod.rs:65-67
        v = self._vectors.get(int(id))
        if v is not None:
            return v
        dim = ctypes.c_size_t()
        if self._L.vdb_hnsw_get_vector(self._h, int(id), None, 0, ctypes.byref(dim)):
            return None
        out = np.zeros(dim.value, dtype=np.float32)
        if self._L.vdb_hnsw_get_vector(self._h, int(id), _fp(out), out.size, ctypes.byref(dim)):
            return None
        return Vector(out)

    def metric(self):
        return self._metric

    def len(self):
        return int(self._L.vdb_hnsw_len(self._h))

    # ---- graph inspection (tests)
    def neighbors(self, id, layer):
        buf = np.zeros(256, dtype=np.uint64)
        n = self._L.vdb_hnsw_neighbors(self._h, int(id), int(layer), _u64p(buf), buf.size)
        return None if n < 0 else [int(x) for x in buf[:n]]

    def level(self, id):
        return int(self._L.vdb_hnsw_node_level(self._h, int(id)))

    def entry_point(self):
        ep, ml = ctypes.c_uint64(), ctypes.c_size_t()
        has = self._L.vdb_hnsw_entry_point(self._h, ctypes.byref(ep), ctypes.byref(ml))
        return (int(ep.value), int(ml.value)) if has else (None, 0)

    def set_traversal(self, host_only=False, host_threads=0):
        """Test hook: host_only=True sends every search through the host traversal (results are identical)."""
        rc = self._L.vdb_hnsw_set_traversal(self._h, int(bool(host_only)), int(host_threads))
        if rc:
            _raise(rc)

    def stats(self):
        out = (ctypes.c_uint64 * 6)()
        self._L.vdb_hnsw_stats(self._h, out)
        return dict(zip(["gpu_distances", "gpu_launches", "last_search_rounds", "last_search_distances", "device_queries",
                         "host_redone"], [int(v) for v in out]))

    def set_build(self, frontier_only=True):
        """Bulk inserts: True (default) = device walks evaluate only what search_layer asks for, the host replays the inserts in
        order; False = the row-scan build (every stored row against every new vector).  The graph is the same."""
        rc = self._L.vdb_hnsw_set_build(self._h, 1 if frontier_only else 0)
        if rc:
            _raise(rc)

    def build_stats(self):
        out = (ctypes.c_uint64 * 8)()
        self._L.vdb_hnsw_build_stats(self._h, out)
        return dict(zip(["frontier_inserts", "walk_distances", "in_chunk_distances", "miss_distances", "miss_round_trips",
                         "record_overflows", "reference_distances", "scan_inserts"], [int(v) for v in out]))

    def build_times(self):
        """Seconds the frontier-only builds spent: mirror sync, waiting for the device walks, host replay, of which misses."""
        out = (ctypes.c_double * 4)()
        self._L.vdb_hnsw_build_times(self._h, out)
        return dict(zip(["mirror_sync_s", "walks_s", "replay_s", "replay_miss_round_trips_s"], [round(float(v), 3) for v in out]))
