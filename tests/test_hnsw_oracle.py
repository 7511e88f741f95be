"""The HNSW oracle (oracle/hnsw_oracle.c) against the reference's own unit tests for this module
(src/hnsw/graph.rs:436-538, src/hnsw/neighbor_queue.rs:150-195, src/hnsw/mod.rs:84-154) and the recall
floors of tests/recall_test.rs:67-80.  CPU only."""
import numpy as np
import pytest

import oracle

EUCLID = 0


def make_graph(seed=1):
    return oracle.HnswOracle(EUCLID, m=4, ef_construction=32, ef_search=16, seed=seed)      # graph.rs:432-434 make_params


def test_heaps_pop_in_neighbor_order():
    # neighbor_queue.rs:154-176: max-heap pops 3,2,1 ; min-heap pops 1,2,3
    d, _ = oracle.heap_replay(+1, [3.0, 1.0, 2.0], [0, 1, 2])
    assert list(d) == [3.0, 2.0, 1.0]
    d, _ = oracle.heap_replay(-1, [3.0, 1.0, 2.0], [0, 1, 2])
    assert list(d) == [1.0, 2.0, 3.0]
    # neighbor_queue.rs:178-189 push_bounded(limit 2) keeps the two closest
    d, _ = oracle.heap_replay(+1, [5.0, 1.0, 3.0], [0, 1, 2], bound=2)
    assert sorted(d) == [1.0, 3.0]
    # ties on distance order by id (neighbor_queue.rs:37-43)
    d, i = oracle.heap_replay(-1, [1.0, 1.0, 1.0, 0.5], [7, 3, 5, 9])
    assert list(i) == [9, 3, 5, 7]


def test_levels_follow_the_reference_formula():
    # graph.rs:118-123: floor(-ln(r) * 1/ln(m)), capped at max_layers - 1
    assert oracle.hnsw_level_from_unit(0.9) == 0
    assert oracle.hnsw_level_from_unit(1.0 / 16 - 1e-9) == 1
    assert oracle.hnsw_level_from_unit(1.0 / 256 - 1e-12) == 2
    assert oracle.hnsw_level_from_unit(0.0) == 15
    assert oracle.hnsw_level_from_unit(1e-300) == 15


def test_insert_single_and_multiple():
    g = make_graph()
    g.insert(0, [1.0, 0.0, 0.0])
    assert len(g) == 1 and g.entry_point()[0] == 0                     # graph.rs:436-443
    g = make_graph()
    for i in range(10):
        g.insert(i, [float(i), 0.0, 0.0])
    assert len(g) == 10                                                 # graph.rs:446-454


def test_self_search():
    # graph.rs:457-485: every inserted vector finds itself at distance ~0
    g = make_graph()
    vecs = [np.array([i * 0.1, (i * 7) * 0.1, (i * 13) * 0.1], dtype=np.float32) for i in range(100)]
    for i, v in enumerate(vecs):
        g.insert(i, v)
    for i, v in enumerate(vecs):
        ids, ds = g.search(v, 1, 16)
        assert len(ids) == 1 and ds[0] < 1e-5, (i, ids, ds)


def test_search_knn():
    # graph.rs:488-504
    g = make_graph()
    for i in range(5):
        g.insert(i, [float(i), 0.0])
    ids, ds = g.search([0.5, 0.0], 2, 16)
    assert len(ids) == 2 and set(ids) == {0, 1}


def test_remove_and_remove_entry_point():
    g = make_graph()
    g.insert(0, [1.0, 0.0])
    g.insert(1, [0.0, 1.0])
    g.remove(0)
    assert len(g) == 1                                                  # graph.rs:507-520
    ids, _ = g.search([0.0, 1.0], 1, 16)
    assert ids[0] == 1
    g.remove(99)                                                        # absent id: Ok(())  (graph.rs:346-348)
    g = make_graph()
    g.insert(0, [1.0, 0.0]); g.insert(1, [0.0, 1.0]); g.insert(2, [1.0, 1.0])
    ep, _ = g.entry_point()
    g.remove(ep)                                                        # graph.rs:523-537
    assert len(g) == 2
    ids, _ = g.search([0.0, 1.0], 1, 16)
    assert len(ids) == 1


def test_hnsw_index_via_trait():
    # mod.rs:88-98 (default params m=16, ef_construction=200; search ef = 50)
    g = oracle.HnswOracle(EUCLID)
    g.insert(0, [1.0, 0.0, 0.0]); g.insert(1, [0.0, 1.0, 0.0]); g.insert(2, [1.0, 1.0, 0.0])
    ids, ds = g.search([1.0, 0.0, 0.0], 2)
    assert len(ids) == 2 and ids[0] == 0 and ds[0] < 1e-5


def test_structure_invariants():
    rng = np.random.default_rng(3)
    g = oracle.HnswOracle(EUCLID, m=8, ef_construction=64, ef_search=32, seed=11)
    rows = rng.random((600, 16), dtype=np.float32)
    for i, v in enumerate(rows):
        g.insert(i, v)
    ep, max_level = g.entry_point()
    assert g.level(ep) == max_level == max(g.level(i) for i in range(600))
    for i in range(600):
        for l in range(g.level(i) + 1):
            nb = g.neighbors(i, l)
            assert len(nb) <= (16 if l == 0 else 8) and len(set(nb)) == len(nb) and i not in nb
            assert all(g.level(j) >= l for j in nb)                      # links only between nodes that exist on the layer
    assert g.neighbors(0, g.level(0) + 1) is None


@pytest.mark.parametrize("n,dim,nq,floor", [(100, 32, 50, 0.90), (1000, 64, 50, 0.90), (5000, 128, 20, 0.85)])
def test_recall_floors_of_the_reference(n, dim, nq, floor):
    # tests/recall_test.rs:28-80: HnswParams::new(16, 200, 50), search_with_ef(k = 10, ef = 100), uniform[0,1) data
    rng = np.random.default_rng(n)
    rows = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((nq, dim), dtype=np.float32)
    g = oracle.HnswOracle(EUCLID, m=16, ef_construction=200, ef_search=50, seed=n)
    for i, v in enumerate(rows):
        g.insert(i, v)
    total = 0.0
    for q in queries:
        truth, _ = oracle.flat_search(EUCLID, rows, q, 10)
        ids, ds = g.search(q, 10, 100)
        assert np.all(ds[1:] >= ds[:-1])
        total += oracle.recall(truth, ids)
    assert total / nq >= floor, total / nq
