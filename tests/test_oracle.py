"""The CPU oracle against the reference's known answers and the committed golden vectors."""
import numpy as np
import pytest

import oracle

M = {"euclidean": oracle.EUCLIDEAN, "cosine": oracle.COSINE, "dot": oracle.DOT}


def test_known_distances(known_answers):
    for c in known_answers["distance"]:
        if c["metric"] == "dot_raw":
            got = -float(oracle.distance(oracle.DOT, c["a"], c["b"]))
        else:
            got = float(oracle.distance(M[c["metric"]], c["a"], c["b"]))
        assert abs(got - c["expect"]) <= c["eps"] * max(1.0, abs(c["expect"])), c["src"]


def test_known_distance_errors(known_answers):
    for c in known_answers["distance_errors"]:
        with pytest.raises(oracle.OracleError) as e:
            oracle.distance(M[c["metric"]], c["a"], c["b"])
        assert e.value.code == oracle.ERR_DIMENSION_MISMATCH, c["src"]


def test_known_norm(known_answers):
    for c in known_answers["norm"]:
        assert abs(oracle.norm(c["x"]) - c["expect"]) <= c["eps"]


def test_known_flat_search(known_answers):
    for c in known_answers["flat_search"]:
        ids = np.array(sorted(int(i) for i in c["rows"]), dtype=np.uint64)
        rows = np.array([c["rows"][str(int(i))] for i in ids], dtype=np.float32)
        got_ids, got_d = oracle.flat_search(M[c["metric"]], rows, c["query"], c["k"], ids=ids)
        assert len(got_ids) == c["expect_len"], c["src"]
        assert int(got_ids[0]) == c["expect_first_id"], c["src"]
        if "expect_first_dist_lt" in c:
            assert got_d[0] < c["expect_first_dist_lt"]
        assert np.all(np.diff(got_d) >= 0)


def test_empty_store(known_answers):
    for c in known_answers["empty_store"]:
        ids, d = oracle.flat_search(M[c["metric"]], np.zeros((0, 3), np.float32), c["query"], c["k"])
        assert len(ids) == 0 and len(d) == 0


def test_filter_cases(known_answers):
    for c in known_answers["filter"]:
        names = sorted(c["rows"])
        rows = np.array([c["rows"][n] for n in names], dtype=np.float32)
        f = c["filter"]
        matches = np.array([c["meta"][n].get(f["field"]) == f["value"] for n in names], dtype=np.uint8)
        ids, _ = oracle.search_with_filter(M[c["metric"]], rows, c["query"], c["k"], matches)
        assert sorted(names[int(i)] for i in ids) == sorted(c["expect_id_set"]), c["src"]


def test_batch_cases(known_answers):
    for c in known_answers["batch"]:
        names = sorted(c["rows"])
        rows = np.array([c["rows"][n] for n in names], dtype=np.float32)
        qs = np.array([q for q, _ in c["queries"]], dtype=np.float32)
        ks = [k for _, k in c["queries"]]
        res = oracle.search_batch(M[c["metric"]], rows, qs, ks)
        assert [[names[int(i)] for i in ids] for ids, _ in res] == c["expect_ids"]


def test_zero_norm_is_invalid_vector():
    rows = np.array([[1, 0], [0, 0]], dtype=np.float32)
    with pytest.raises(oracle.OracleError) as e:
        oracle.flat_search(oracle.COSINE, rows, [1, 1], 1)
    assert e.value.code == oracle.ERR_INVALID_VECTOR
    with pytest.raises(oracle.OracleError):
        oracle.flat_search(oracle.COSINE, rows[:1], [0, 0], 1)
    # tiny values whose squares underflow to zero also give norm == 0.0 (vector.rs:35-37)
    with pytest.raises(oracle.OracleError):
        oracle.flat_search(oracle.COSINE, np.array([[1e-30, 1e-30]], np.float32), [1, 1], 1)
    oracle.flat_search(oracle.EUCLIDEAN, rows, [0, 0], 2)


def test_sequential_fold_order():
    # 1 + 2^-24 * 2 (twice) differs between a left fold and a pairwise sum
    a = np.array([1.0, 2.0 ** -24, 2.0 ** -24, 2.0 ** -24], dtype=np.float32)
    one = np.ones(4, dtype=np.float32)
    got = -float(oracle.distance(oracle.DOT, a, one))
    s = np.float32(0)
    for x in a:
        s = np.float32(s + x)
    assert got == float(s)
    assert got != float(np.float32(np.float64(a).sum()))


def test_k_larger_than_n_and_k_zero():
    rows = np.eye(3, dtype=np.float32)
    ids, d = oracle.flat_search(oracle.EUCLIDEAN, rows, [1, 0, 0], 10)
    assert list(ids) == [0, 1, 2]
    ids, d = oracle.flat_search(oracle.EUCLIDEAN, rows, [1, 0, 0], 0)
    assert len(ids) == 0


def test_golden_vectors_replay(golden_cases):
    g = golden_cases
    names = sorted({k.split("/")[0] for k in g.files})
    assert names
    for name in names:
        rows, queries, ids = g[f"{name}/rows"], g[f"{name}/queries"], g[f"{name}/ids"]
        for mname, m in M.items():
            for k in g[f"{name}/ks"]:
                key = f"{name}/{mname}/k{k}/ids"
                if key not in g.files:
                    continue
                for b in range(queries.shape[0]):
                    i, d = oracle.flat_search(m, rows, queries[b], int(k), ids=ids)
                    assert np.array_equal(i, g[key][b])
                    assert np.array_equal(d, g[f"{name}/{mname}/k{k}/dists"][b])


def test_recall_definition():
    assert oracle.recall([1, 2, 3, 4], [4, 3, 9, 8]) == 0.5
