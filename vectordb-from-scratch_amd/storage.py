"""Host-side mirror of the reference's VectorStore (src/storage.rs:83-348): String id <-> internal
id maps, metadata, dimension enforcement, the metadata filter, and the batch drivers that call the
index.  Same names, argument meaning and error behaviour as the reference; `search_batch` hands the
whole batch to the index in one call (the hook SURVEY.md 8(b) describes)."""
from dataclasses import dataclass, field

import numpy as np

from .error import DimensionMismatch, VectorNotFound
from .index import GpuFlatIndex, Index
from .vector import DistanceMetric, Vector


@dataclass
class SearchResult:                                  # storage.rs:13-16
    id: str
    distance: float


class Metadata:                                      # storage.rs:19-42
    def __init__(self, fields=None):
        self._fields = dict(fields or {})

    def insert(self, key, value):
        self._fields[key] = value

    def get(self, key):
        return self._fields.get(key)

    def fields(self):
        return self._fields


class MetadataFilter:                                # storage.rs:45-71
    def __init__(self, op, field=None, value=None, filters=None):
        self.op, self.field, self.value, self.filters = op, field, value, filters or []

    @staticmethod
    def Eq(field, value):
        return MetadataFilter("eq", field, value)

    @staticmethod
    def Ne(field, value):
        return MetadataFilter("ne", field, value)

    @staticmethod
    def Exists(field):
        return MetadataFilter("exists", field)

    @staticmethod
    def And(filters):
        return MetadataFilter("and", filters=filters)

    @staticmethod
    def Or(filters):
        return MetadataFilter("or", filters=filters)

    def matches(self, metadata):                     # storage.rs:62-70
        if self.op == "eq":
            return metadata.get(self.field) == self.value and metadata.get(self.field) is not None
        if self.op == "ne":
            return metadata.get(self.field) != self.value
        if self.op == "exists":
            return metadata.get(self.field) is not None
        if self.op == "and":
            return all(f.matches(metadata) for f in self.filters)
        if self.op == "or":
            return any(f.matches(metadata) for f in self.filters)
        raise ValueError(self.op)


@dataclass
class BatchInsertItem:                               # storage.rs:74-79
    id: str
    vector: Vector
    metadata: Metadata = field(default_factory=Metadata)


class VectorStore:
    """VectorStore<I: Index>  (storage.rs:83-95).  `VectorStore(metric)` builds the GPU flat index
    where the reference's `VectorStore::new` builds a FlatIndex (storage.rs:99-101)."""

    def __init__(self, metric=None, index=None, device=0):
        if index is None:
            index = GpuFlatIndex(DistanceMetric(metric), device=device)
        assert isinstance(index, Index)
        self._index = index
        self._id_to_internal = {}
        self._internal_to_id = {}
        self._metadata = {}
        self._next_id = 0
        self._dimension = None

    @classmethod
    def with_index(cls, index):                      # storage.rs:118-127
        return cls(index=index)

    # ---- mutation
    def insert(self, id, vector):                    # storage.rs:130-132
        self.insert_with_metadata(id, vector, Metadata())

    def insert_with_metadata(self, id, vector, metadata):   # storage.rs:135-172
        id = str(id)
        dim = vector.dimension()
        if self._dimension is not None:
            if dim != self._dimension:
                raise DimensionMismatch(self._dimension, dim)
        else:
            self._dimension = dim
        old = self._id_to_internal.get(id)
        if old is not None:
            self._index.remove(old)
            self._metadata.pop(old, None)
            self._internal_to_id.pop(old, None)
        internal = self._next_id
        self._next_id += 1
        self._index.add(internal, vector)
        self._id_to_internal[id] = internal
        self._internal_to_id[internal] = id
        self._metadata[internal] = metadata

    def insert_batch(self, items):                   # storage.rs:293-298
        for it in items:
            self.insert_with_metadata(it.id, it.vector, it.metadata)

    def delete(self, id):                            # storage.rs:175-192
        internal = self._id_to_internal.pop(id, None)
        if internal is None:
            raise VectorNotFound(id)
        v = self._index.get_vector(internal)
        if v is None:
            v = Vector([])
        self._internal_to_id.pop(internal, None)
        self._metadata.pop(internal, None)
        self._index.remove(internal)
        return v

    # ---- reads
    def get(self, id):                               # storage.rs:195-198
        internal = self._id_to_internal.get(id)
        return None if internal is None else self._index.get_vector(internal)

    def get_metadata(self, id):                      # storage.rs:201-204
        internal = self._id_to_internal.get(id)
        return None if internal is None else self._metadata.get(internal)

    def len(self):
        return self._index.len()

    def __len__(self):
        return self.len()

    def is_empty(self):
        return self._index.is_empty()

    def metric(self):
        return self._index.metric()

    def dimension(self):
        return self._dimension

    def index(self):
        return self._index

    def list_ids(self):
        return list(self._id_to_internal)

    def _check_dim(self, query):
        if self._dimension is not None and query.dimension() != self._dimension:
            raise DimensionMismatch(self._dimension, query.dimension())

    def _map(self, index_results):
        out = []
        for internal, dist in index_results:         # storage.rs:234-242
            sid = self._internal_to_id.get(internal)
            if sid is not None:
                out.append(SearchResult(sid, float(dist)))
        return out

    # ---- search
    def search(self, query, k):                      # storage.rs:217-245
        if self.is_empty():
            return []
        self._check_dim(query)
        return self._map(self._index.search(query, k))

    def _post_filter(self, index_results, k, flt):   # storage.rs:272-287
        out = []
        for internal, dist in index_results:
            sid = self._internal_to_id.get(internal)
            meta = self._metadata.get(internal)
            if sid is None or meta is None:
                continue
            if flt.matches(meta):
                out.append(SearchResult(sid, float(dist)))
                if len(out) == k:
                    break
        return out

    def search_with_filter(self, query, k, flt):     # storage.rs:249-290  (post-filter, 3x over-fetch)
        if self.is_empty():
            return []
        self._check_dim(query)
        fetch_k = min(max(k * 3, k), self.len())
        return self._post_filter(self._index.search(query, fetch_k), k, flt)

    def search_batch(self, queries):                 # storage.rs:302-310, one index call
        if self.is_empty():
            return [[] for _ in queries]
        for q, _ in queries:
            self._check_dim(q)
        return [self._map(r) for r in self._index.search_batch(list(queries))]

    def search_batch_with_filter(self, queries, flt):   # storage.rs:313-322
        if self.is_empty():
            return [[] for _ in queries]
        for q, _ in queries:
            self._check_dim(q)
        n = self.len()
        fetch = [(q, min(max(k * 3, k), n)) for q, k in queries]
        res = self._index.search_batch(fetch)
        return [self._post_filter(r, k, flt) for r, (_, k) in zip(res, queries)]

    # ---- BASELINE config 4: the filter compiled to a device bitmask applied BEFORE top-k.
    # The reference's post-filter result is always a prefix of this one.
    def compile_filter(self, flt):
        bits = max(self._next_id, 1)
        mask = np.zeros((bits + 63) // 64, dtype=np.uint64)
        for internal, meta in self._metadata.items():
            if flt.matches(meta):
                mask[internal >> 6] |= np.uint64(1) << np.uint64(internal & 63)
        return mask, bits

    def search_batch_prefiltered(self, queries, flt):
        if self.is_empty():
            return [[] for _ in queries]
        for q, _ in queries:
            self._check_dim(q)
        mask, bits = self.compile_filter(flt)
        return [self._map(r) for r in self._index.search_batch(list(queries), id_mask=mask, mask_bits=bits)]
