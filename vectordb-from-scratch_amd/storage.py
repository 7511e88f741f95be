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


class _Column:
    """One metadata field as a dictionary-encoded column over internal ids: codes[internal] = index into `values`
    (-1: the row has no such field).  What `HashMap<usize, Metadata>` (storage.rs:90) holds per row, transposed so that a
    filter over a million rows is a few numpy passes instead of a million dict lookups."""

    def __init__(self):
        self.codes = np.full(1024, -1, dtype=np.int32)
        self.values, self.code_of = [], {}

    def _grow(self, n):
        if n > self.codes.size:
            new = np.full(max(n, 2 * self.codes.size), -1, dtype=np.int32)
            new[:self.codes.size] = self.codes
            self.codes = new

    def code(self, value, create=False):
        value = str(value)                               # Metadata is HashMap<String, String> (storage.rs:19-22): one encoding for set and set_range
        c = self.code_of.get(value)
        if c is None and create:
            c = self.code_of[value] = len(self.values)
            self.values.append(value)
        return c

    def set(self, internal, value):
        self._grow(internal + 1)
        self.codes[internal] = self.code(value, create=True)

    def set_range(self, start, values):
        """values: a sequence of n strings (or None for 'no such field') for internal ids start .. start+n-1."""
        vals = np.asarray(values, dtype=object)
        self._grow(start + vals.size)
        uniq, inv = np.unique(np.where(vals == None, "", vals).astype(str), return_inverse=True)   # noqa: E711
        lut = np.array([self.code(u, create=True) for u in uniq], dtype=np.int32)
        codes = lut[inv]
        if (vals == None).any():                                                                  # noqa: E711
            codes = np.where(vals == None, -1, codes)                                             # noqa: E711
        self.codes[start:start + vals.size] = codes

    def view(self, n):
        self._grow(n)
        return self.codes[:n]


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
        # filter compilation: the metadata as dictionary-encoded columns + a presence bitmap, kept in
        # step with every insert / upsert / delete
        self._cols = {}
        self._present = np.zeros(1024, dtype=bool)
        self._bulk = []                                  # attach_bulk_metadata ranges: (first internal id, n, ids or None)

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
        if old is None:
            old = self._bulk_internal(id)                # an id that names a bulk-attached row: the insert replaces that row
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
        self._mark_present(internal, True)
        for key, value in metadata.fields().items():
            self._cols.setdefault(key, _Column()).set(internal, value)
        if old is not None:
            self._mark_present(old, False)

    def insert_batch(self, items):                   # storage.rs:293-298
        for it in items:
            self.insert_with_metadata(it.id, it.vector, it.metadata)

    def delete(self, id):                            # storage.rs:175-192
        internal = self._id_to_internal.pop(id, None)
        if internal is None:
            internal = self._bulk_internal(id)
        if internal is None:
            raise VectorNotFound(id)
        v = self._index.get_vector(internal)
        if v is None:
            v = Vector([])
        self._internal_to_id.pop(internal, None)
        self._metadata.pop(internal, None)
        self._index.remove(internal)
        self._mark_present(internal, False)
        return v

    # ---- reads
    def _internal_of(self, id):
        """String id -> internal id, for inserted rows and for bulk-attached rows alike (None: unknown)."""
        internal = self._id_to_internal.get(id)
        return internal if internal is not None else self._bulk_internal(id)

    def _metadata_of(self, internal):
        """The row's Metadata (storage.rs:90): the stored object of an inserted row, or the columns of a bulk-attached row
        materialised on demand."""
        meta = self._metadata.get(internal)
        if meta is None and self._bulk_id(internal) is not None:
            meta = Metadata({k: c.values[c.codes[internal]] for k, c in self._cols.items()
                             if internal < c.codes.size and c.codes[internal] >= 0})
        return meta

    def get(self, id):                               # storage.rs:195-198
        internal = self._internal_of(id)
        return None if internal is None else self._index.get_vector(internal)

    def get_metadata(self, id):                      # storage.rs:201-204
        internal = self._internal_of(id)
        return None if internal is None else self._metadata_of(internal)

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
        out = list(self._id_to_internal)
        for start, n, ids, _ in self._bulk:
            live = np.nonzero(self._present[start:start + n])[0]
            out.extend(str(start + int(j)) if ids is None else ids[int(j)] for j in live)
        return out

    def _mark_present(self, internal, on):
        if internal >= self._present.size:
            new = np.zeros(max(internal + 1, 2 * self._present.size), dtype=bool)
            new[:self._present.size] = self._present
            self._present = new
        self._present[internal] = on

    def attach_bulk_metadata(self, n, columns, ids=None):
        """Register n rows that are ALREADY in the index under the next n internal ids (a bulk device load, a mapped
        vector file: persistence/mmap.rs has no id or metadata column) together with their metadata columns
        {field: sequence of n strings / None}.  String ids default to the decimal internal id.  The per-row
        `Metadata` objects of the reference (storage.rs:90) are materialised only when asked for (get_metadata)."""
        n = int(n)
        start = self._next_id
        if ids is not None and len(ids) != n:
            raise ValueError("ids must have n entries")
        # one id, one row: a bulk id that names a row already in the store would make upsert / delete act on the wrong one
        new_ids = [str(i) for i in ids] if ids is not None else None
        if new_ids is not None:
            if len(set(new_ids)) != n:
                raise ValueError("bulk ids must be distinct")
            clash = next((i for i in new_ids if self._internal_of(i) is not None), None)
        else:
            clash = next((i for i in self._id_to_internal if i.isdigit() and str(int(i)) == i and start <= int(i) < start + n), None)
            if clash is None:
                clash = next((i for s0, m, other, _ in self._bulk if other is not None
                              for i in other if i.isdigit() and str(int(i)) == i and start <= int(i) < start + n), None)
        if clash is not None:
            raise ValueError(f"bulk id {clash!r} is already in use")
        for key, values in columns.items():
            if len(values) != n:
                raise ValueError(f"column {key!r} must have n entries")
            self._cols.setdefault(key, _Column()).set_range(start, values)
        self._mark_present(start + n - 1, False)                 # grow once
        self._present[start:start + n] = True
        self._bulk.append((start, n, new_ids, None if new_ids is None else {sid: start + j for j, sid in enumerate(new_ids)}))
        self._next_id += n
        if self._dimension is None and hasattr(self._index, "dim"):
            self._dimension = self._index.dim() or None
        return start

    def _bulk_id(self, internal):
        for start, n, ids, _ in self._bulk:
            if start <= internal < start + n:
                if not self._present[internal]:
                    return None
                return str(internal) if ids is None else ids[internal - start]
        return None

    def _bulk_internal(self, id):
        """Internal id of a LIVE bulk-attached row named `id` (None: no such row)."""
        id = str(id)
        for start, n, ids, lut in self._bulk:
            if ids is None:
                if id.isdigit() and str(int(id)) == id and start <= int(id) < start + n and self._present[int(id)]:
                    return int(id)
            else:
                internal = lut.get(id)
                if internal is not None and self._present[internal]:
                    return internal
        return None

    def _check_dim(self, query):
        if self._dimension is not None and query.dimension() != self._dimension:
            raise DimensionMismatch(self._dimension, query.dimension())

    def _map(self, index_results):
        out = []
        for internal, dist in index_results:         # storage.rs:234-242
            sid = self._internal_to_id.get(internal)
            if sid is None and self._bulk:
                sid = self._bulk_id(internal)
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
            if sid is None and self._bulk:
                sid = self._bulk_id(internal)             # rows registered by attach_bulk_metadata live in the columns only
            meta = self._metadata_of(internal) if sid is not None else None
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
    def _eval_filter(self, flt, n):
        """MetadataFilter::matches (storage.rs:60-71) for ALL internal ids below n at once: a bool array."""
        if flt.op in ("eq", "ne", "exists"):
            col = self._cols.get(flt.field)
            if col is None:                                       # nobody has the field: Eq / Exists never match, Ne always does
                return np.full(n, flt.op == "ne", dtype=bool)
            codes = col.view(n)
            if flt.op == "exists":
                return codes >= 0
            c = col.code(flt.value) if flt.value is not None else None
            if flt.op == "eq":
                return (codes == c) if c is not None else np.zeros(n, dtype=bool)
            return (codes != c) if c is not None else np.ones(n, dtype=bool)      # Ne: a missing field matches (storage.rs:65)
        if flt.op in ("and", "or"):
            acc = np.full(n, flt.op == "and", dtype=bool)          # all([]) is true, any([]) is false, like Rust's iterators
            for f in flt.filters:
                m = self._eval_filter(f, n)
                acc = (acc & m) if flt.op == "and" else (acc | m)
            return acc
        raise ValueError(flt.op)

    def compile_filter(self, flt):
        """The filter as the id bitmask the device applies before top-k (bit i of word i>>6 = internal id i eligible).
        Vectorised over the dictionary-encoded metadata columns: eq / ne / exists are one comparison of an int32 column,
        and / or combine bool arrays, the presence bitmap removes deleted and superseded ids, np.packbits packs."""
        bits = max(self._next_id, 1)
        m = self._eval_filter(flt, bits)
        pres = self._present[:bits] if self._present.size >= bits else np.concatenate([self._present, np.zeros(bits - self._present.size, dtype=bool)])
        m = m & pres
        words = (bits + 63) // 64
        packed = np.zeros(words * 8, dtype=np.uint8)
        pb = np.packbits(m, bitorder="little")
        packed[:pb.size] = pb
        return packed.view(np.uint64), bits

    def search_batch_prefiltered(self, queries, flt):
        if self.is_empty():
            return [[] for _ in queries]
        for q, _ in queries:
            self._check_dim(q)
        mask, bits = self.compile_filter(flt)
        return [self._map(r) for r in self._index.search_batch(list(queries), id_mask=mask, mask_bits=bits)]
