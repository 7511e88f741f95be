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


def _searchable_store(vdb, order):
    """A store over an Index whose search returns the live ids in a scripted order (distance = rank)."""
    class ScriptedIndex(vdb.Index):
        def __init__(self):
            self.live = set()

        def add(self, id, vector):
            self.live.add(id)

        def remove(self, id):
            self.live.discard(id)

        def search(self, query, k):
            return [(i, float(r)) for r, i in enumerate(j for j in order if j in self.live)][:k]

        def get_vector(self, id):
            return vdb.Vector([float(id)]) if id in self.live else None

        def metric(self):
            return vdb.DistanceMetric.Euclidean

        def len(self):
            return len(self.live)

        def dim(self):
            return 1

    st = vdb.VectorStore.with_index(ScriptedIndex())
    return st


def test_post_filter_sees_bulk_attached_rows():
    """`_post_filter` looked ids and metadata up in the per-row maps only, so every row registered by
    attach_bulk_metadata was dropped by search_with_filter / search_batch_with_filter -- the server's default filtered route."""
    vdb = load_package()
    F = vdb.MetadataFilter
    n = 12
    st = _searchable_store(vdb, order=list(range(n)))
    st.index().live.update(range(n))                               # the rows are already in the index (a bulk device load)
    colors = np.array(["red", "green", "blue"], dtype=object)[np.arange(n) % 3]
    st.attach_bulk_metadata(n, {"color": colors})
    q = vdb.Vector([0.0])
    assert [r.id for r in st.search(q, 4)] == ["0", "1", "2", "3"]
    # storage.rs:249-290: over-fetch 3k, keep the matches in rank order, take k
    got = st.search_with_filter(q, 2, F.Eq("color", "red"))
    assert [r.id for r in got] == ["0", "3"]
    assert [[r.id for r in rs] for rs in st.search_batch_with_filter([(q, 1), (q, 3)], F.Eq("color", "green"))] == [["1"], ["1", "4", "7"]]
    # the reference's result is a prefix of the pre-filter (bitmask) result
    mask, bits = st.compile_filter(F.Eq("color", "red"))
    assert [i for i in range(bits) if (int(mask[i >> 6]) >> (i & 63)) & 1] == [0, 3, 6, 9]
    # a mixed store: inserted rows and bulk rows answer the same filter together
    st.insert_with_metadata("late", vdb.Vector([99.0]), vdb.Metadata({"color": "red"}))
    st.index().search = lambda query, k, ix=st.index(): ([(n, -1.0)] + [(i, float(i)) for i in range(n) if i in ix.live])[:k]
    assert [r.id for r in st.search_with_filter(q, 3, F.Eq("color", "red"))] == ["late", "0", "3"]


def test_bulk_rows_are_first_class_get_delete_list_upsert_and_collisions():
    """get / delete / list_ids / get_metadata resolve bulk ids (default decimal AND custom), an insert under a
    bulk row's id replaces that row, colliding ids are refused, and column values are strings whichever way they arrived."""
    import pytest
    vdb = load_package()
    st = _searchable_store(vdb, order=list(range(100)))
    st.insert("5", vdb.Vector([1.0]))                              # a user id that LOOKS like a default bulk id
    st.index().live.update(range(1, 5))
    with pytest.raises(ValueError):
        st.attach_bulk_metadata(6, {"c": ["a"] * 6})                # internal ids 1..6 -> default ids "1".."6": "5" is taken
    st.attach_bulk_metadata(4, {"c": ["a", "b", 7, None]}, ids=["w", "x", "y", "z"])
    with pytest.raises(ValueError):
        st.attach_bulk_metadata(1, {"c": ["a"]}, ids=["x"])
    assert sorted(st.list_ids()) == ["5", "w", "x", "y", "z"]
    assert st.get("x").data[0] == 2.0 and st.get("nope") is None
    assert st.get_metadata("y").fields() == {"c": "7"} and st.get_metadata("z").fields() == {}
    assert st.get_metadata("w").get("c") == "a"
    st.insert_with_metadata("k", vdb.Vector([3.0]), vdb.Metadata({"c": 7}))      # set() and set_range() encode alike
    mask, bits = st.compile_filter(vdb.MetadataFilter.Eq("c", "7"))
    assert [i for i in range(bits) if (int(mask[i >> 6]) >> (i & 63)) & 1] == [3, 5]
    st.delete("x")
    assert st.get("x") is None and "x" not in st.list_ids() and 2 not in st.index().live
    with pytest.raises(vdb.VectorNotFound):
        st.delete("x")
    st.insert_with_metadata("w", vdb.Vector([9.0]), vdb.Metadata({"c": "new"}))   # upsert of a bulk row: the old row goes
    assert 1 not in st.index().live and st.get_metadata("w").get("c") == "new" and sorted(st.list_ids()) == ["5", "k", "w", "y", "z"]
