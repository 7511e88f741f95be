// vdb_cert.cpp -- the coefficients of the "certified top-k" error bounds and the plans of the tiers that use them
// (DESIGN.md 4.1).  Everything here is host arithmetic on index constants; the tests of these bounds are
// tests/test_gpu_certificate.py.
#include <cmath>

#include "vdb_index.h"

namespace vdbi {

// Certification coefficient (DESIGN.md "certified top-k"): worst-case rounding bound of the
// MFMA fma chain plus the oracle's sequential fold, K = padded row length.  VDB_EPS_SCALE
// scales it (diagnostics only).
float eps_coef(const Index* ix) {
    const double u = 5.9604644775390625e-08;   // 2^-24
    double K = (double)ix->ld;
    double c;
    if (ix->metric == vdb::EUCLID) c = (K + 4.0) * u;
    else if (ix->metric == vdb::COSINE) c = (2.0 * K + 16.0) * u;
    else c = (2.0 * K + 2.0) * u;
    c *= 1.1;
    c *= ix->kn.eps_scale;                 // 1.0 outside the diagnostics build
    return (float)c;
}

// f32 accumulation inside the bf16 MFMAs (products of two bf16 are exact in f32): at most K 2^-22 |q||d|, 5 % margin.
// The operand-rounding part of the screening tier's error bound is evaluated per query in rerank_kernel from the
// known |q - bf16(q)| and the per-index maxima of |d - bf16(d)| (row_stats_kernel).
float c_acc_bf16(const Index* ix) {
    double c = (double)ix->ld * 2.384185791015625e-07 * 1.05;
    c *= ix->kn.eps_scale;                 // 1.0 outside the diagnostics build
    return (float)c;
}

// The LOCAL form of the screening tier's certificate (Dot / Euclid).  For a row d and a query q the ranking score differs
// from what the oracle's exact distance implies by at most
//     Dot:     1.01 (|e_q||d| + 1.004 |q||e_d|) + (c_acc + eps) |q||d|
//     Euclid:  2 x that with eps doubled, + eps |d|^2 (folded into the row's beta), + eps (|q|^2 + e_k^2) (query only, cert_test)
// with e_q = q - bf16(q), e_d = d - bf16(d).  Everything row-dependent is of the form |q| A_d + |e_q| B_d;
// with kappa = B_d / A_d of a typical row (relative bf16 rounding error 1e-3) it is bounded by g_q * M_d,
//     g_q = |q| + kappa |e_q|   (query_prep),      M_d = max(A_d, B_d / kappa)   (row_stats: ONE more constant per row),
// and the kernels rank by the lower-bound score  score - g_q M_d.  Any kappa > 0 is valid; this one makes the bound tight.
MarginPlan margin_plan(const Index* ix) {
    MarginPlan mp;
    if (ix->metric == vdb::COSINE) return mp;
    const double eps = (double)eps_coef(ix), cacc = (double)c_acc_bf16(ix);
    const double two = ix->metric == vdb::EUCLID ? 2.0 : 1.0;
    const double Ae = two * 1.01 * 1.004, An = two * (cacc + eps), Bn = two * 1.01;
    const double kappa = Bn / (Ae * 1.0e-3 + An);
    mp.m_e = (float)Ae; mp.m_n = (float)(An * 1.000001); mp.m_b = (float)(Bn / kappa * 1.000001); mp.kappa = (float)(kappa * 1.000001);
    mp.beta_shrink = ix->metric == vdb::EUCLID ? (float)eps : 0.0f;
    return mp;
}

// bf16 screening tier: the select delivers up to 256 candidates per query, sorted by score, and the re-rank goes
// through them adaptively (rerank_kernel): first round_up(k + 22, 32), then 32 more per round until the result is
// certified.  The filter threshold is the kt-th smallest of the M = S/64 group minima of an S-row sample (at least
// kt rows pass it); S = 2^s is sized so that about 2000 keys per query pass, and kt <= M/4 so that the kt smallest
// minima come from (nearly) distinct groups.
Bf16Plan plan_bf16(const vdb_flat_index* ix, uint32_t n, size_t k) {
    Bf16Plan pl;
    if (n < BF16_MIN_ROWS || k > 112) return pl;
    // threshold rank: at least kt rows pass the filter, about kt * n / S are expected to (k = 10 at 1M rows: 16 -> ~244 keys per
    // query).  The re-rank certifies against the score of the first candidate it did NOT re-rank, so what the rank has to
    // provide is a pool a few times deeper than the first round (k + 38), not a margin: 16 instead of 32 halves the appends
    // of the filter pass (its epilogue's rare path, ~12 us per launch at config 2) at the same first-round certification
    const uint32_t want_kt = std::min<uint32_t>(128u, round_up((uint32_t)k + 6u, 16u));
    uint64_t S = std::min<uint64_t>(65536u, std::max<uint64_t>(16384u, pow2_ceil((uint64_t)n / 16u)));
    if (ix->kn.sample16) S = pow2_ceil(std::max(256u, ix->kn.sample16));
    while (S / 256u < want_kt && 2 * S <= n / 2) S *= 2;
    while (S > n) S /= 2;
    // threshold rank: enough for the first re-rank round; the pool (about N/S * kt keys) feeds the deeper rounds
    uint32_t kt = std::min<uint32_t>(want_kt, (uint32_t)(S / 256u));
    // ... and no deeper than that: one sample rank stands for n / S rows, so rank kt lets about kt * n / S keys through.  Aim at
    // ~6 k keys per query (k = 100 at 1.25M rows: rank 32 -> ~610 keys instead of rank 112 -> ~2100, of which the select kept
    // 512 anyway).  Below rank k + 1 the sample no longer GUARANTEES k candidates; it does not have to -- the re-rank refuses
    // to certify a result with fewer than k real rows and the query goes to the re-threshold pass (never observed: the pool
    // size varies by about +-18 % at rank 32).
    {
        const uint64_t per_rank = std::max<uint64_t>(1, (uint64_t)n / S);
        const uint32_t kt_pool = round_up((uint32_t)std::max<uint64_t>(16, (6ull * k + per_rank - 1) / per_rank), 16u);
        kt = std::min(kt, kt_pool);
    }
    // ... and no shallower than four first re-rank rounds (k + 38 rows each) of expected pool: below 250k rows the sample cannot
    // shrink with the index any more (S = 16384), one sample rank stands for fewer rows, and rank 16 let only ~120 keys per query
    // through at 125k rows -- 3 in 10000 (query, shard) pairs then ran out of pool before their certificate closed, which at
    // eight row shards of a 1M-row index sent every second BATCH into the re-threshold pass and a second exchange.  With the
    // floor (125k rows: rank 32) none in 15000 did, at the same step time.
    {
        const double per_rank = (double)n / (double)S;
        const uint32_t kt_floor = round_up((uint32_t)std::ceil(4.0 * ((double)k + 38.0) / per_rank), 8u);
        kt = std::min<uint32_t>(std::max(kt, kt_floor), (uint32_t)(S / 256u));
    }
    if (ix->kn.kt16) kt = std::min<uint32_t>(ix->kn.kt16, (uint32_t)(S / 256u));
    if (kt < 16) return pl;
    pl.kp = k > 48 ? 512 : 256;                                  // candidates the select delivers (depth limit of the re-rank)
    pl.S = (uint32_t)S; pl.kt = kt;
    while ((1ull << pl.shift) < S) ++pl.shift;
    return pl;
}

uint32_t pick_kp(size_t k) {
    size_t want = k + std::max<size_t>(6, k / 5);
    if (want <= 32) return 32;
    if (want <= 64) return 64;
    if (want <= 128) return 128;
    return 0;   // exact-scan path
}

}  // namespace vdbi
