"""The screening tier's CERTIFICATE, tested directly (VERDICT r1 "what's weak" 2, "next round" 1a).

The default search ranks rows by bf16-MFMA scores and returns the reference's exact f32 distances (distance.rs:37-73)
only because a bound on the score error proves that no excluded row can enter the top k.  End-to-end parity on benign
data cannot tell a loose-but-wrong bound from a right one, so this file checks the bound itself, on adversarial inputs,
through the diagnostics entry points of include/vdb_flat.h:

  * the scores come from the PRODUCTION filter kernel (vdb_flat_debug_screen_scores: thresholds open, every row kept);
  * (A) soundness by the PRODUCTION certification function: for every (query, row) pair, cert_test(T = the row's own
    score, e_k = the row's own exact distance) must be false -- a certificate may never exclude a row at its own distance;
  * (B) the same in numbers: |what the score implies - the oracle's exact distance| <= the error budget the certificate
    grants, evaluated in float64 with the documented formulas; the worst ratio is recorded;
  * (C) the MFMA accumulation term alone: |acc - fp64 dot of the bf16-rounded operands| <= c_acc |q||d|.

Worst observed ratios are written to gpurun_out/cert_margins.json (copied to profiles/ per round).
"""
import json
import os
import zlib

import numpy as np
import pytest

import oracle
from conftest import ROOT, load_package

pytestmark = pytest.mark.gpu

MARGINS = {}


@pytest.fixture(scope="module")
def vdb():
    v = load_package()
    v.build()
    yield v
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and MARGINS:
        with open(os.path.join(out, "cert_margins.json"), "w") as f:
            json.dump(MARGINS, f, indent=1, sort_keys=True)


def bf16_rne(x):
    """RNE rounding of f32 to bf16 (what v_cvt_pk_bf16_f32 keeps), as f32."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def make_case(kind, metric, rng):
    """(rows, queries): adversarial for the error bound."""
    if kind == "cancel":
        # heavy cancellation: alternating signs, every dot product is a difference of two large nearly equal sums
        n, d, nq = 66_000, 96, 48
        base = rng.random((n, d), dtype=np.float32) + 0.5
        sign = np.where(np.arange(d) % 2 == 0, 1.0, -1.0).astype(np.float32)
        rows = base * sign
        q = (rng.random((nq, d), dtype=np.float32) + 0.5) * sign * np.where(rng.random((nq, 1)) < 0.5, 1, -1).astype(np.float32)
        q[: nq // 2] = np.abs(q[: nq // 2])                      # these cancel against the rows' signs
    elif kind == "scales":
        # magnitudes from 1e-21 to 1e+17 between rows AND inside a row (1e18 squared times the dimension overflows f32)
        n, d, nq = 66_100, 64, 48
        top = 17.0                                               # |d|^2 (and Euclid's |d|^2 + 2|q||d|) must stay finite in f32
        rexp = rng.uniform(-18, top, (n, 1))
        rows = (rng.standard_normal((n, d)) * 10.0 ** (rexp + rng.uniform(-3, 0, (n, d)))).astype(np.float32)
        qexp = rng.uniform(-18, top, (nq, 1))
        q = (rng.standard_normal((nq, d)) * 10.0 ** (qexp + rng.uniform(-3, 0, (nq, d)))).astype(np.float32)
    elif kind == "dim16384":
        n, d, nq = 1_500, 16384, 8
        rows = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q[0] = rows[7] + 1e-4 * rng.standard_normal(d).astype(np.float32)
    elif kind == "bf16ties":
        # clusters whose bf16 images collide: perturbations far below the bf16 resolution of the values
        n, d, nq = 65_800, 128, 32
        centres = rng.standard_normal((n // 200 + 1, d)).astype(np.float32)
        rows = (centres[np.arange(n) // 200] * (1.0 + 2e-5 * rng.standard_normal((n, d)))).astype(np.float32)
        q = (centres[rng.integers(0, centres.shape[0], nq)] * (1.0 + 1e-5 * rng.standard_normal((nq, d)))).astype(np.float32)
    elif kind == "subnormal":
        # f32 subnormals (below 1.18e-38) and values whose products underflow
        n, d, nq = 65_700, 40, 24
        rows = (rng.standard_normal((n, d)) * 10.0 ** rng.uniform(-41, -17, (n, 1))).astype(np.float32)
        q = (rng.standard_normal((nq, d)) * 10.0 ** rng.uniform(-24, -15, (nq, 1))).astype(np.float32)
        q[::2] = (rng.standard_normal((nq // 2, d)) * 10.0 ** rng.uniform(-6, 2, (nq // 2, 1))).astype(np.float32)   # ordinary queries against tiny rows
        rows[:8] = (rng.standard_normal((8, d)) * 1e-3).astype(np.float32)   # a few ordinary rows keep Cosine defined
    else:
        raise ValueError(kind)
    if metric == 1:
        rows[np.linalg.norm(rows.astype(np.float64), axis=1) == 0] = 1.0
        bad = ~np.isfinite(np.linalg.norm(rows.astype(np.float64), axis=1).astype(np.float32)) | \
            (np.sqrt((rows.astype(np.float32) ** 2).sum(1)) == 0)
        rows[bad] = 1.0                                         # a zero (or f32-underflowing) norm fails every Cosine search
        qb = np.sqrt((q.astype(np.float32) ** 2).sum(1)) == 0
        q[qb] = 1.0
    return rows, q


def exact_distances(metric, rows, q):
    """Oracle distances of every row (f32 bits of the reference's arithmetic), as [n] f32, via one full-k search."""
    n = rows.shape[0]
    ids, d = oracle.flat_search(metric, rows, q, n)
    out = np.empty(n, dtype=np.float32)
    out[ids.astype(np.int64)] = d
    return out


@pytest.mark.parametrize("kind", ["cancel", "scales", "dim16384", "bf16ties", "subnormal"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_score_error_stays_inside_the_certified_bound(vdb, metric, kind):
    rng = np.random.default_rng(zlib.crc32(f"{metric}/{kind}".encode()))
    rows, q = make_case(kind, metric, rng)
    n, dim = rows.shape
    nq = q.shape[0]
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    ix.add_bulk(rows)
    s_raw, qinfo, c = ix.debug_screen_scores(q, raw=True)
    s_lb, qinfo2, c2 = ix.debug_screen_scores(q, raw=False)                      # leaves the prepared queries for the probes
    info = ix.debug_row_info().astype(np.float64)
    nd, margin = info[:, 0], info[:, 3]
    assert s_raw.shape == (nq, n) and np.array_equal(qinfo, qinfo2)
    assert c2["lower_bound_scores"] == (0.0 if metric == 1 else 1.0)
    assert not np.isnan(s_raw).any() and not np.isnan(s_lb).any(), "a live row got no key"
    eps, c_acc, ld = c["eps_coef"], c["c_acc"], c["ld"]
    R64, Q64 = rows.astype(np.float64), q.astype(np.float64)
    qn_true, nd_true = np.linalg.norm(Q64, axis=1), np.linalg.norm(R64, axis=1)
    worst = {"A_violations": 0, "B_ratio": 0.0, "C_ratio": 0.0}
    TINY = 2.0 ** -40
    nd_pos_min = float(nd[nd > 0].min())
    uncertifiable = 0
    D16 = bf16_rne(q).astype(np.float64) @ bf16_rne(rows).astype(np.float64).T   # [nq, n]
    for b in range(nq):
        dist = exact_distances(metric, rows, q[b])
        assert np.isfinite(dist).all(), "test data must give finite distances"
        qn, eq, g = (float(x) for x in qinfo[b, :3])
        # ---- (A) the production certification function on (own score, own distance)
        fl = ix.debug_cert_probe(np.full(n, b, np.uint32), s_lb[b], dist)
        worst["A_violations"] += int(fl.sum())
        assert fl.sum() == 0, (metric, kind, b, "cert_test would exclude a row at its own distance", np.nonzero(fl)[0][:5])
        # ---- (B) error against budget, float64.  Outside the certifiable domain (kernels_aux.hip "UNDERFLOW": a query
        # norm below 2^-40, or under Cosine an index holding a row that small) cert_test refuses outright -- checked by (A)
        # above, which ran for every pair -- and there is no budget to compare with.
        if qn < TINY or (metric == 1 and nd_pos_min < TINY):
            uncertifiable += 1
            continue
        s = s_raw[b].astype(np.float64)
        d64 = dist.astype(np.float64)
        floor = ld * 2.0 ** -140
        if metric == 2:       # Dot: dist = -fold_dot; score = -acc
            err = np.abs(s - d64)
            budget = g * margin + floor
        elif metric == 0:     # Euclid: dist^2 ~ score + |q|^2 ; beta was shrunk by eps |d|^2 on top of the margin
            err = np.abs(s + qn * qn - d64 * d64)
            budget = g * margin + eps * nd * nd + eps * (qn * qn + d64 * d64) + 4 * floor
        else:                 # Cosine: dist ~ 1 + score / |q|
            err = np.abs(1.0 + s / qn - d64)
            budget = 1.01 * (eq / qn + 1.004 * c["rho_max"]) + c_acc + eps + floor / (qn * TINY)
        ok = budget > 0
        ratio = np.max(err[ok] / budget[ok]) if np.ndim(budget) else float(np.max(err) / budget)
        assert np.all(err[~ok] == 0) if np.ndim(budget) else True
        worst["B_ratio"] = max(worst["B_ratio"], float(ratio))
        # the lower-bound score really is score - g * margin (one fma)
        if metric != 1:
            lb = s_lb[b].astype(np.float64)
            fin = np.isfinite(lb) & np.isfinite(margin)
            assert np.all(lb[fin] <= s[fin] - g * margin[fin] * (1 - 1e-5) + np.abs(s[fin]) * 2.4e-7 + 1e-42)
        # ---- (C) f32 accumulation inside the bf16 MFMAs (Dot: score = -acc exactly)
        if metric == 2:
            acc = -s
            den = c_acc * qn_true[b] * nd_true + floor          # (products that underflow f32 are covered by the absolute floor)
            okc = den > 0
            worst["C_ratio"] = max(worst["C_ratio"], float(np.max(np.abs(acc - D16[b])[okc] / den[okc])))
    MARGINS[f"{['euclid', 'cosine', 'dot'][metric]}/{kind}"] = {**worst, "queries_outside_the_certifiable_domain": uncertifiable, "rows": n, "dim": dim, "queries": nq, "eps_coef": eps, "c_acc": c_acc}
    print(f"\n[certificate] metric={metric} {kind}: worst error/budget = {worst['B_ratio']:.4f}"
          + (f", worst MFMA accumulation error / (c_acc |q||d|) = {worst['C_ratio']:.4f}" if metric == 2 else ""))
    if kind in ("cancel", "dim16384", "bf16ties"):
        assert uncertifiable == 0
    assert worst["B_ratio"] <= 1.0, worst
    assert worst["C_ratio"] <= 1.0, worst


@pytest.mark.parametrize("kind", ["cancel", "scales", "bf16ties"])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_f32_tier_score_error_stays_inside_its_bound(vdb, metric, kind):
    """The same for the f32-input-MFMA tier (`eps_coef` "is similarly asserted"): its scores
    (v_mfma_f32_32x32x2_f32 fma chain, raw = 2) against the oracle's distances, budget = eps_coef terms only."""
    rng = np.random.default_rng(zlib.crc32(f"f32/{metric}/{kind}".encode()))
    rows, q = make_case(kind, metric, rng)
    rows, q = np.ascontiguousarray(rows[:20_000]), np.ascontiguousarray(q[:16])
    n = rows.shape[0]
    ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
    ix.add_bulk(rows)
    s32, qinfo, c = ix.debug_screen_scores(q, raw=2)
    info = ix.debug_row_info().astype(np.float64)
    nd = info[:, 0]
    eps, ld, ndmax = c["eps_coef"], c["ld"], c["nd_max"]
    assert not np.isnan(s32).any()
    TINY, floor = 2.0 ** -40, ld * 2.0 ** -140
    nd_pos_min = float(nd[nd > 0].min())
    worst = 0.0
    for b in range(q.shape[0]):
        dist = exact_distances(metric, rows, q[b])
        qn = float(qinfo[b, 0])
        fl = ix.debug_cert_probe(np.full(n, b, np.uint32), s32[b], dist)
        assert fl.sum() == 0, (metric, kind, b, np.nonzero(fl)[0][:5])
        if qn < TINY or (metric == 1 and nd_pos_min < TINY):
            continue
        s, d64 = s32[b].astype(np.float64), dist.astype(np.float64)
        if metric == 2:
            err, budget = np.abs(s - d64), eps * qn * ndmax + floor
        elif metric == 0:      # (beta carries -eps |d|^2 since round 2: the safe direction, added to the two-sided budget)
            err, budget = np.abs(s + qn * qn - d64 * d64), eps * ((qn + ndmax) ** 2 + d64 * d64) + eps * nd * nd + 4 * floor
        else:
            err, budget = np.abs(1.0 + s / qn - d64), eps + floor / (qn * TINY)
        worst = max(worst, float(np.max(err / budget)))
    MARGINS[f"f32_tier/{['euclid', 'cosine', 'dot'][metric]}/{kind}"] = {"B_ratio": worst, "rows": n, "dim": rows.shape[1], "eps_coef": eps}
    print(f"\n[certificate] f32 tier metric={metric} {kind}: worst error/budget = {worst:.4f}")
    assert worst <= 1.0


def test_adversarial_data_still_gives_the_oracle_top_k(vdb):
    """The same adversarial sets through the ordinary search: certified answers equal the oracle's, whichever tier gave them."""
    for metric in (0, 1, 2):
        for kind in ("cancel", "scales", "bf16ties", "subnormal"):
            rng = np.random.default_rng(zlib.crc32(f"{metric}/{kind}/1".encode()))
            rows, q = make_case(kind, metric, rng)
            ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
            ix.add_bulk(rows)
            gi, gd, gc = ix.search_batch_arrays(q, 10)
            st = ix.last_stats()
            assert st["bf16_screen"] == 1, st
            for b in range(0, q.shape[0], 5):
                oi, od = oracle.flat_search(metric, rows, q[b], 10)
                assert gc[b] == 10 and np.array_equal(gi[b], oi) and np.array_equal(gd[b].view(np.uint32), od.view(np.uint32)), (metric, kind, b, st)


def test_one_huge_norm_row_loosens_only_its_own_certificate(vdb):
    """VERDICT r1 next-round 5 / DESIGN 9.3: per-index maxima let ONE 1e6-norm row loosen every Dot / Euclid certificate and
    push tight clusters to the slower tiers.  With per-row margins the outlier costs nothing: every query is certified
    by the screening tier (no f32-tier, re-threshold or exact-scan queries), results bit-identical to the oracle."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 200_000, 96, 64, 10
    centres = rng.standard_normal((n // 400, d)).astype(np.float32)
    rows = (centres[np.arange(n) // 400] + 0.05 * rng.standard_normal((n, d))).astype(np.float32)    # tight clusters, cluster-ordered
    rows[123_456] = (rng.standard_normal(d) * 1e6 / np.sqrt(d)).astype(np.float32)                   # the outlier: norm 1e6
    q = (rows[rng.integers(0, n, nq)] + 0.02 * rng.standard_normal((nq, d))).astype(np.float32)
    for metric in (2, 0):
        ix = vdb.GpuFlatIndex(vdb.DistanceMetric(metric), keep_host_copy=False)
        ix.add_bulk(rows)
        gi, gd, gc = ix.search_batch_arrays(q, k)
        st = ix.last_stats()
        assert st["bf16_screen"] == 1 and st["f32_tier_queries"] == 0 and st["exact_queries"] == 0, st
        for b in range(0, nq, 9):
            oi, od = oracle.flat_search(metric, rows, q[b], k)
            assert np.array_equal(gi[b], oi) and np.array_equal(gd[b].view(np.uint32), od.view(np.uint32)), (metric, b)
