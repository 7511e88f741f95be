"""Host logic of the VectorStore mirror that needs no GPU: filter compilation.

`compile_filter` turns a MetadataFilter (src/storage.rs:47-71) into the id bitmask the device applies before top-k.  It
is vectorised over dictionary-encoded metadata columns; this file checks it against the reference's own per-row
`MetadataFilter::matches` on stores that went through inserts, upserts, deletes and a bulk attach."""
import numpy as np

from conftest import load_package


def _store(vdb):
    class FakeIndex(vdb.Index):                                  # the Index trait with no device behind it
        def __init__(self):
            self.rows = {}

        def add(self, id, vector):
            self.rows[id] = vector

        def remove(self, id):
            self.rows.pop(id, None)

        def search(self, query, k):
            return []

        def get_vector(self, id):
            return self.rows.get(id)

        def metric(self):
            return vdb.DistanceMetric.Euclidean

        def len(self):
            return len(self.rows)

    return vdb.VectorStore.with_index(FakeIndex())


def _ids(mask, bits):
    return [i for i in range(bits) if (int(mask[i >> 6]) >> (i & 63)) & 1]


def test_compile_filter_equals_per_row_matches_after_upserts_and_deletes():
    vdb = load_package()
    F, M, V = vdb.MetadataFilter, vdb.Metadata, vdb.Vector
    rng = np.random.default_rng(3)
    st = _store(vdb)
    colors, sizes = ["red", "green", "blue"], ["s", "m", "l", "xl"]
    for step in range(3000):
        sid = f"v{rng.integers(0, 900)}"                           # collisions: upserts (a new internal id each time, storage.rs:157-164)
        md = {}
        if rng.random() < 0.8:
            md["color"] = colors[rng.integers(0, 3)]
        if rng.random() < 0.5:
            md["size"] = sizes[rng.integers(0, 4)]
        st.insert_with_metadata(sid, V([float(step)]), M(md))
        if rng.random() < 0.1 and st.list_ids():
            st.delete(st.list_ids()[rng.integers(0, len(st.list_ids()))])
    filters = [F.Eq("color", "red"), F.Eq("color", "nope"), F.Ne("color", "green"), F.Ne("color", "nope"), F.Ne("missing", "x"),
               F.Exists("size"), F.Exists("missing"), F.And([F.Eq("color", "red"), F.Eq("size", "l")]),
               F.Or([F.Eq("color", "green"), F.Eq("size", "s")]), F.And([F.Ne("color", "red"), F.Or([F.Exists("size"), F.Eq("color", "blue")])]),
               F.And([]), F.Or([])]
    for f in filters:
        mask, bits = st.compile_filter(f)
        want = sorted(i for i, md in st._metadata.items() if f.matches(md))      # storage.rs:60-71, row by row
        assert bits == st._next_id and _ids(mask, bits) == want, f.op


def test_bulk_attached_columns_and_string_ids():
    """attach_bulk_metadata registers rows that are already in the index (bulk device load / mapped vector file) with their
    metadata as columns; compile_filter sees them, results map back to string ids, get_metadata rebuilds the row's map."""
    vdb = load_package()
    F = vdb.MetadataFilter
    st = _store(vdb)
    st.insert_with_metadata("first", vdb.Vector([1.0, 2.0]), vdb.Metadata({"color": "red"}))
    n = 10_000
    palette = np.array(["red", "green", "blue", "amber"], dtype=object)
    col = palette[np.arange(n) % 4]
    col[7] = None                                                  # a row without the field
    start = st.attach_bulk_metadata(n, {"color": col})
    assert start == 1 and st._next_id == n + 1
    mask, bits = st.compile_filter(F.Eq("color", "red"))
    got = _ids(mask, bits)
    assert got == [0] + [start + i for i in range(n) if i % 4 == 0]
    mask, bits = st.compile_filter(F.Exists("color"))
    assert len(_ids(mask, bits)) == n                              # the seeded row + n - 1 bulk rows
    mask, bits = st.compile_filter(F.Ne("color", "red"))
    assert (start + 7) in _ids(mask, bits)                         # Ne matches a missing field (storage.rs:65)
    assert st._map([(start + 5, 0.5), (0, 0.7)]) == [vdb.SearchResult(str(start + 5), 0.5), vdb.SearchResult("first", 0.7)]
    assert st.get_metadata(str(start + 6)).fields() == {"color": "blue"}
    assert st.get_metadata(str(start + 7)).fields() == {}
