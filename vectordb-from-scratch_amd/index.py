"""Host-side mirror of the reference's `Index` trait (src/index.rs:11-35) and the GPU-backed
implementation that replaces `FlatIndex` (src/flat_index.rs:12-74) behind it.

`GpuFlatIndex` owns ids and a host copy of each row (the trait's `get_vector` returns a
borrow, so the host keeps one anyway) and calls the hand-written HIP kernels through the C
ABI of include/vdb_flat.h.  There is no CPU search path."""
import abc
import ctypes
import os

import numpy as np

from . import _ffi
from .error import DimensionMismatch, IndexError_, InvalidVector, NanDistance
from .vector import DistanceMetric, Vector


def _raise(rc):
    msg, exp, act = _ffi.last_error()
    if rc == _ffi.ERR_DIMENSION_MISMATCH:
        raise DimensionMismatch(exp, act)
    if rc == _ffi.ERR_INVALID_VECTOR:
        raise InvalidVector(msg.split(": ", 1)[-1])
    if rc == _ffi.ERR_NAN:
        raise NanDistance(msg)
    raise IndexError_(msg)


class Index(abc.ABC):
    """trait Index  (src/index.rs:11-35)"""

    @abc.abstractmethod
    def add(self, id, vector): ...

    @abc.abstractmethod
    def remove(self, id): ...

    @abc.abstractmethod
    def search(self, query, k): ...

    @abc.abstractmethod
    def get_vector(self, id): ...

    @abc.abstractmethod
    def metric(self): ...

    @abc.abstractmethod
    def len(self): ...

    def is_empty(self):                              # index.rs:32-34
        return self.len() == 0

    def search_batch(self, queries):
        """The provided method SURVEY.md 8(b) proposes adding to the trait: default = the
        sequential loop of VectorStore::search_batch (storage.rs:306-309)."""
        return [self.search(q, k) for q, k in queries]

    def __len__(self):
        return self.len()


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class GpuFlatIndex(Index):
    """Drop-in for FlatIndex (src/flat_index.rs) backed by the MI355X engine."""

    EXCHANGE_RCCL, EXCHANGE_PEER = 0, 1

    def __init__(self, metric, device=0, keep_host_copy=True, devices=None):
        """devices=[d0, d1, ...]: ONE index whose rows are sharded over these GPUs inside this process
        (vdb_flat_create_sharded; queries and outputs live on d0).  Every method below works on it unchanged."""
        self._metric = DistanceMetric(metric)
        self._h = ctypes.c_void_p()
        self._L = _ffi.lib()
        if devices is not None:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._L.vdb_flat_create_sharded(int(self._metric), devs, len(devices), ctypes.byref(self._h))
            device = devices[0] if len(devices) else 0
        else:
            rc = self._L.vdb_flat_create(int(self._metric), int(device), ctypes.byref(self._h))
        if rc:
            _raise(rc)
        self._device = int(device)
        self._keep = keep_host_copy
        self._vectors = {}                           # id -> Vector (for get_vector borrows)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.vdb_flat_destroy(h)

    # ---- Index trait
    def add(self, id, vector):                       # flat_index.rs:38-41
        v = vector if isinstance(vector, Vector) else Vector(vector)
        rc = self._L.vdb_flat_add(self._h, int(id), _fp(v.data), v.dimension())
        if rc:
            _raise(rc)
        if self._keep:
            self._vectors[int(id)] = v

    def remove(self, id):                            # flat_index.rs:43-46
        rc = self._L.vdb_flat_remove(self._h, int(id))
        if rc:
            _raise(rc)
        self._vectors.pop(int(id), None)

    def get_vector(self, id):                        # flat_index.rs:48-50
        if self._keep:
            return self._vectors.get(int(id))
        dim = ctypes.c_size_t()
        rc = self._L.vdb_flat_get_vector(self._h, int(id), None, 0, ctypes.byref(dim))
        if rc == _ffi.ERR_NOT_FOUND:
            return None
        if rc:
            _raise(rc)
        out = np.zeros(dim.value, dtype=np.float32)
        rc = self._L.vdb_flat_get_vector(self._h, int(id), _fp(out), out.size, ctypes.byref(dim))
        if rc:
            _raise(rc)
        return Vector(out)

    def search(self, query, k):                      # flat_index.rs:52-65
        q = query if isinstance(query, Vector) else Vector(query)
        res = self.search_batch([(q, k)])
        return res[0]

    def metric(self):                                # flat_index.rs:67-69
        return self._metric

    def len(self):                                   # flat_index.rs:71-73
        return int(self._L.vdb_flat_len(self._h))

    # ---- batched hot call (overrides the provided loop)
    def search_batch(self, queries, id_mask=None, mask_bits=0):
        """queries: sequence of (Vector, k).  Returns a list of [(id, distance), ...] per query."""
        if len(queries) == 0:
            return []
        dims = {q.dimension() for q, _ in queries}
        if len(dims) != 1:
            # a ragged batch: the reference handles each query on its own (storage.rs:306-309)
            return [self.search_batch([(q, k)], id_mask, mask_bits)[0] for q, k in queries]
        qs = np.stack([q.data for q, _ in queries]).astype(np.float32, copy=False)
        ks = np.array([int(k) for _, k in queries], dtype=np.uintp)
        ids, dists, counts = self.search_batch_arrays(qs, ks, id_mask=id_mask, mask_bits=mask_bits)
        return [[(int(ids[b, i]), np.float32(dists[b, i])) for i in range(int(counts[b]))] for b in range(len(queries))]

    def search_batch_arrays(self, queries, k, id_mask=None, mask_bits=0):
        """numpy in/out form: queries [nq, dim] f32, k an int or a per-query array.
        Returns (ids u64 [nq,kmax], dists f32 [nq,kmax], counts [nq])."""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = qs.shape
        if np.isscalar(k):
            ks_ptr, kscalar, kmax = None, int(k), int(k)
        else:
            ks = np.ascontiguousarray(k, dtype=np.uintp)
            ks_ptr = ks.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t))
            kscalar, kmax = 0, int(ks.max()) if ks.size else 0
        kstride = max(kmax, 1)
        out_ids = np.zeros((nq, kstride), dtype=np.uint64)
        out_d = np.zeros((nq, kstride), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uintp)
        mask_ptr = None
        if id_mask is not None:
            m = np.ascontiguousarray(id_mask, dtype=np.uint64)
            mask_ptr = _u64p(m)
        rc = self._L.vdb_flat_search_batch(self._h, _fp(qs), nq, dim, ks_ptr, kscalar, mask_ptr, int(mask_bits),
                                           kstride, _u64p(out_ids), _fp(out_d),
                                           counts.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)))
        if rc:
            _raise(rc)
        return out_ids, out_d, counts

    # ---- bulk build and device-resident entry points
    def add_bulk(self, rows, ids=None, first_id=0):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n, dim = rows.shape
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            idp = _u64p(ids)
        rc = self._L.vdb_flat_add_bulk(self._h, idp, int(first_id), _fp(rows), n, dim)
        if rc:
            _raise(rc)
        if self._keep:
            for i in range(n):
                self._vectors[int(ids[i]) if ids is not None else first_id + i] = Vector(rows[i])

    def add_bulk_device(self, dev_ptr, n, dim, ids=None, first_id=0):
        """rows already in this GPU's HBM (e.g. a torch tensor's data_ptr()); no host copy is kept."""
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint64)
            idp = _u64p(ids)
        rc = self._L.vdb_flat_add_bulk_device(self._h, idp, int(first_id), ctypes.c_void_p(dev_ptr), int(n), int(dim))
        if rc:
            _raise(rc)

    def load_vector_file(self, path, first_id=0):
        """Bulk-load the reference's mmap vector file (src/persistence/mmap.rs); returns the row count."""
        n = ctypes.c_size_t()
        rc = self._L.vdb_flat_load_vector_file(self._h, os.fsencode(path), int(first_id), ctypes.byref(n))
        if rc:
            _raise(rc)
        return n.value

    def reserve(self, rows, dim):
        rc = self._L.vdb_flat_reserve(self._h, int(rows), int(dim))
        if rc:
            _raise(rc)

    def flush(self):
        rc = self._L.vdb_flat_flush(self._h)
        if rc:
            _raise(rc)

    def search_batch_device(self, q_ptr, nq, dim, k, out_ids_ptr, out_dists_ptr, out_counts_ptr, stream=0,
                            mask_ptr=0, mask_bits=0):
        """Everything resident in HBM: raw device pointers (uint64 ids, f32 dists, u32 counts)."""
        rc = self._L.vdb_flat_search_batch_device(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(dim), int(k), ctypes.c_void_p(mask_ptr or None),
            int(mask_bits), ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dists_ptr),
            ctypes.c_void_p(out_counts_ptr), ctypes.c_void_p(stream or None))
        if rc:
            _raise(rc)

    def search_batch_device_begin(self, q_ptr, nq, dim, k, out_ids_ptr, out_dists_ptr, out_counts_ptr, code_ptr=0,
                                  stream=0, mask_ptr=0, mask_bits=0):
        """First tier enqueued, no host synchronisation; *code_ptr (device int32) = 0 or VDB_PENDING_HOST (100).
        Must be followed by search_batch_device_finish() from the same thread."""
        rc = self._L.vdb_flat_search_batch_device_begin(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(dim), int(k), ctypes.c_void_p(mask_ptr or None),
            int(mask_bits), ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dists_ptr),
            ctypes.c_void_p(out_counts_ptr), ctypes.c_void_p(code_ptr or None), ctypes.c_void_p(stream or None))
        if rc:
            _raise(rc)

    def search_batch_device_finish(self):
        """Waits, runs the fallback tiers where needed; returns True when outputs were rewritten."""
        changed = ctypes.c_int(0)
        rc = self._L.vdb_flat_search_batch_device_finish(self._h, ctypes.byref(changed))
        if rc:
            _raise(rc)
        return bool(changed.value)

    def search_batch_device_submit(self, q_ptr, nq, dim, k, out_ids_ptr, out_dists_ptr, out_counts_ptr, stream=0,
                                   mask_ptr=0, mask_bits=0):
        """Asynchronous form: the first tier is enqueued, a ticket comes back at once; at most two tickets per handle."""
        t = ctypes.c_int(-1)
        rc = self._L.vdb_flat_search_batch_device_submit(
            self._h, ctypes.c_void_p(q_ptr), int(nq), int(dim), int(k), ctypes.c_void_p(mask_ptr or None),
            int(mask_bits), ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dists_ptr),
            ctypes.c_void_p(out_counts_ptr), ctypes.c_void_p(stream or None), ctypes.byref(t))
        if rc:
            _raise(rc)
        return t.value

    def search_batch_device_wait(self, ticket):
        """Waits for a submitted search, runs its fallback tiers where needed; its outputs are complete on return."""
        rc = self._L.vdb_flat_search_batch_device_wait(self._h, int(ticket))
        if rc:
            _raise(rc)

    def distances_batch(self, queries, id_lists):
        """Exact reference distances of query b to the stored ids id_lists[b] (HNSW candidate lists)."""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = qs.shape
        offsets = np.zeros(nq + 1, dtype=np.uintp)
        offsets[1:] = np.cumsum([len(l) for l in id_lists])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.uint64) for l in id_lists])
                                   if int(offsets[-1]) else np.zeros(0, np.uint64))
        out = np.zeros(int(offsets[-1]), dtype=np.float32)
        rc = self._L.vdb_flat_distances_batch(self._h, _fp(qs), nq, dim,
                                              offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), _u64p(ids), _fp(out))
        if rc:
            _raise(rc)
        return [out[int(offsets[b]):int(offsets[b + 1])] for b in range(nq)]

    def last_stats(self):
        out = (ctypes.c_uint64 * 16)()
        self._L.vdb_flat_last_stats_ex(self._h, out, 16)
        keys = ["mfma_queries", "exact_queries", "pool_overflows", "rows_scanned", "sample_rows", "kprime",
                "uncertified", "fused_kernel_ns", "bf16_screen", "f32_tier_queries", "host_enqueued_ns",
                "host_flags_ns", "host_total_ns", "rethreshold_queries", "shadow_rows", "diag_knobs_active"]
        return dict(zip(keys, [int(v) for v in out]))

    def set_screen(self, mode):
        """1 (default): bf16 screening tier first; 0: f32 MFMA tier only.  Results are identical."""
        rc = self._L.vdb_flat_set_screen(self._h, int(mode))
        if rc:
            _raise(rc)

    def set_wide(self, on=True):
        """Batches above 256 queries: 512 queries per fetch of the rows (default) or 256.  Results are identical."""
        rc = self._L.vdb_flat_set_wide(self._h, 1 if on else 0)
        if rc:
            _raise(rc)

    TIERS_NO_RETHRESHOLD, TIERS_FORCE_F32, TIERS_FORCE_EXACT, TIERS_NO_DIRECT = 1, 2, 4, 8

    def set_shadow(self, on=True):
        """Opt-in bf16 shadow of the rows for the screening pass (include/vdb_flat.h: +50 % device memory, half the HBM bytes
        per batch, results identical).  last_stats()["shadow_rows"] tells whether the last search used it."""
        rc = self._L.vdb_flat_set_shadow(self._h, 1 if on else 0)
        if rc != 0:
            _raise(rc)

    def set_sample_cache(self, on=True):
        """The screening tier's compact bf16 copy of its sample rows (include/vdb_flat.h; on by default, results identical)."""
        rc = self._L.vdb_flat_set_sample_cache(self._h, 1 if on else 0)
        if rc != 0:
            _raise(rc)

    def set_tiers(self, flags):
        """Test hook: force the hand-over of queries to the slower tiers (VDB_TIERS_*).  Results are identical."""
        rc = self._L.vdb_flat_set_tiers(self._h, int(flags))
        if rc:
            _raise(rc)

    # ---- certificate diagnostics (include/vdb_flat.h "Diagnostics of the screening tier's CERTIFICATE")
    def debug_screen_scores(self, queries, raw=False):
        """(scores [nq, rows] f32, qinfo [nq, 4], consts dict) from the production filter kernel with open thresholds.
        raw: 0 the scores the screening tier ranks by, 1 its plain scores, 2 the f32 MFMA tier's scores."""
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = qs.shape
        n = int(self._L.vdb_flat_debug_rows(self._h))
        scores = np.empty((nq, n), dtype=np.float32)
        qinfo = np.zeros((nq, 4), dtype=np.float32)
        consts = np.zeros(8, dtype=np.float64)
        rc = self._L.vdb_flat_debug_screen_scores(self._h, _fp(qs), nq, dim, int(raw), _fp(scores), _fp(qinfo),
                                                  consts.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        if rc:
            _raise(rc)
        keys = ["eps_coef", "c_acc", "kappa", "nd_max", "ed_max", "rho_max", "lower_bound_scores", "ld"]
        return scores, qinfo, dict(zip(keys, consts.tolist()))

    def debug_last_thresholds(self, nq):
        """The screening tier's per-query filter thresholds of the last search (first nq queries)."""
        out = np.zeros(int(nq), dtype=np.float32)
        rc = self._L.vdb_flat_debug_last_thresholds(self._h, _fp(out), int(nq))
        if rc:
            _raise(rc)
        return out

    def debug_row_info(self):
        """[rows, 4] f32: exact-order norm, alpha, beta, margin."""
        n = int(self._L.vdb_flat_debug_rows(self._h))
        out = np.zeros((n, 4), dtype=np.float32)
        rc = self._L.vdb_flat_debug_row_info(self._h, _fp(out), n)
        if rc:
            _raise(rc)
        return out

    def debug_cert_probe(self, qi, T, ek):
        """The production certification test for (prepared query qi[i], score bound T[i], k-th exact distance ek[i])."""
        qi = np.ascontiguousarray(qi, dtype=np.uint32)
        T = np.ascontiguousarray(T, dtype=np.float32)
        ek = np.ascontiguousarray(ek, dtype=np.float32)
        out = np.zeros(qi.size, dtype=np.uint32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        rc = self._L.vdb_flat_debug_cert_probe(self._h, qi.ctypes.data_as(u32p), _fp(T), _fp(ek), qi.size, out.ctypes.data_as(u32p))
        if rc:
            _raise(rc)
        return out

    def set_profile(self, on=True):
        rc = self._L.vdb_flat_set_profile(self._h, int(bool(on)))
        if rc:
            _raise(rc)

    def dim(self):
        return int(self._L.vdb_flat_dim(self._h))

    # ---- sharded handles (vdb_flat_create_sharded)
    def shards(self):
        return int(self._L.vdb_flat_shards(self._h))

    def shard_len(self, shard):
        return int(self._L.vdb_flat_shard_len(self._h, int(shard)))

    def set_exchange(self, mode):
        """EXCHANGE_RCCL (grouped ncclAllGather over in-process communicators) or EXCHANGE_PEER (peer copies into devices[0])."""
        rc = self._L.vdb_flat_set_exchange(self._h, int(mode))
        if rc:
            _raise(rc)

    def shard_stats(self):
        out = (ctypes.c_uint64 * 8)()
        rc = self._L.vdb_flat_shard_stats(self._h, out)
        if rc:
            _raise(rc)
        keys = ["exchanges", "shards", "exchange_mode", "rccl_ranks", "host_total_ns", "host_enqueued_ns"]
        return dict(zip(keys, [int(v) for v in out]))
