// vdb_search.cpp -- the tier scheduler of the batched search: bf16 screening pass, re-threshold pass,
// f32-input MFMA tier, exact scan.  Each tier either PROVES its answer is the oracle's or hands the query on; results never
// depend on which tier answered.  Reference: FlatIndex::search (src/flat_index.rs:52-65) driven by
// VectorStore::search_batch (src/storage.rs:302-310).
#include <cmath>
#include <cstdio>
#include <limits>

#include "vdb_index.h"

namespace vdbi {

// The filter pass of the screening tier: over the bf16 shadow rows when the index keeps them (vdb_flat_set_shadow) and the
// row pitch allows whole 128-byte row requests, else over the f32 rows.  Same scores either way, bit for bit.
bool shadow_usable(const Index* ix) { return ix->d_rows16 && ix->ld % 64 == 0; }
void launch_filter_pass(Index* ix, vdb::FusedBf16Params& fp, hipStream_t s) {
    if (shadow_usable(ix)) { fp.rows16 = ix->d_rows16; vdb::launch_fused_s16(fp, s); return; }
#ifdef VDB_DIAG
    if (!ix->kn.fused_pipe) { vdb::launch_fused_bf16(fp, s); return; }
#endif
    vdb::launch_fused_bf16p(fp, s);
}

// ------------------------------------------------------------------ exact path for one query
int exact_one(Index* ix, hipStream_t s, uint32_t q, size_t k, const uint32_t* d_rowmask, uint64_t* d_out_ids,
              float* d_out_dists, uint32_t* d_out_count) {
    int rc;
    uint32_t n = ix->n_uploaded;
    if ((rc = ensure_ranks(ix))) return rc;
    if ((rc = ix->cur->w_exact.ensure(n))) return rc;
    if ((rc = ix->cur->w_exsel.ensure(MAX_SELECT + 8))) return rc;
    if ((rc = ix->cur->w_cnt.ensure(4 * SUPER + 16))) return rc;
    vdb::ExactScanParams ep{ix->d_rows, ix->ld, ix->dim, n, ix->cur->w_qp.p + (size_t)q * ix->ld, ix->cur->w_qnorm.p + q, ix->d_nd,
                            d_rowmask, ix->ids_monotone ? nullptr : ix->d_idrank.p, ix->metric, ix->cur->w_exact.p,
                            ix->cur->w_flags.p};
    vdb::launch_exact_scan(ep, s);
    uint32_t* cnt = ix->cur->w_cnt.p + 4 * SUPER;
    uint64_t* last = ix->cur->w_exsel.p + MAX_SELECT;      // largest key emitted so far (one u64 after the sort area)
    // k may be as large as the index: emit in chunks of MAX_SELECT, each chunk = the smallest keys
    // strictly above the previous chunk's last key
    for (size_t done = 0; done < k; done += MAX_SELECT) {
        uint32_t kk = (uint32_t)std::min<size_t>(MAX_SELECT, k - done);
        vdb::SelectParams sp{};
        sp.keys = ix->cur->w_exact.p; sp.stride = 0; sp.counts = nullptr; sp.n_fixed = n; sp.cap = n;
        sp.kk = kk; sp.out_keys = ix->cur->w_exsel.p; sp.out_stride = MAX_SELECT; sp.out_cnt = cnt;
        sp.out_thr = nullptr; sp.ovf = nullptr;
        sp.lo_excl = done ? last : nullptr; sp.out_last = last;
        vdb::launch_select(sp, 1, s);
        vdb::EmitParams em{ix->cur->w_exsel.p, MAX_SELECT, cnt, ix->ids_monotone ? nullptr : ix->d_rank2row.p,
                           ix->d_row_ids, d_out_ids + done, d_out_dists + done, d_out_count, kk, done ? 1u : 0u};
        vdb::launch_emit(em, s);
    }
    HIP_TRY(hipGetLastError());
    return VDB_OK;
}

// ------------------------------------------------------------------ tier: f32 MFMA scores + certified re-rank
// Runs the f32 pipeline for the nq queries whose padded rows start at qp (stride ld; the block
// must be readable and zero up to a multiple of 256 rows, thr = -inf in the padding), writing results for query j
// at out_*[j*k ..] and the certification / pool-overflow flags at d_cert[j] / d_ovf[j].
int pass_f32(Index* ix, hipStream_t s, const float* qp, const float* qnorm, float* thr, uint32_t nq, size_t k, uint32_t kp,
             const uint32_t* d_rowmask, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_cert,
             uint32_t* d_ovf, uint32_t* d_status) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const bool small = n <= SMALL_N;
    // Threshold sample size S: the fused pass keeps about n*kp/S keys per query, spread over 512 private
    // sub-pools of 64 slots and gathered into 16384 LDS slots by the select.  S is chosen so that this
    // expectation stays near 8000 or below (mean sub-pool fill <= 16), and the sample costs <= ~3 % of the
    // fused pass for k = 10.
    uint32_t S = n;
    if (!small) {
        uint64_t want = std::max<uint64_t>(n / 256u, (uint64_t)n * kp / 8000u);
        S = (uint32_t)std::min<uint64_t>(65536u, std::max<uint64_t>(2048u, pow2_ceil(want)));
        if (ix->kn.sample) S = std::min<uint32_t>(n, std::max(64u, ix->kn.sample));
    }
    // candidate pools: one private sub-pool per (query, row range, row part, lane half) of the fused kernel
    const uint32_t capl = 64;
    // sub-pools in one pass = queries * row ranges * row parts * 2 = 512 * n_cu for every kernel shape
    const size_t pass_subs = 512u * (size_t)ix->n_cu;
    if ((rc = ix->cur->w_dense.ensure((size_t)SUPER * S))) return rc;
    if ((rc = ix->cur->w_cand.ensure((size_t)SUPER * kp))) return rc;
    if (!small) {
        if ((rc = ix->cur->w_samp.ensure((size_t)SUPER * kp))) return rc;
        if ((rc = ix->cur->w_pool.ensure(pass_subs * capl))) return rc;
        if ((rc = ix->cur->w_subcnt.ensure(pass_subs))) return rc;
    }
    uint32_t* d_cnt_a = ix->cur->w_cnt.p;               // sample select counts
    uint32_t* d_cand_cnt = ix->cur->w_cnt.p + 2 * SUPER;
    if (!ix->cur->stats[8]) { ix->cur->stats[4] = S; ix->cur->stats[5] = kp; }
    const float eps = eps_coef(ix);

    for (uint32_t q0 = 0; q0 < nq; q0 += SUPER) {
        const uint32_t nb = std::min(SUPER, nq - q0);
        const uint32_t tiles = (nb + 31) / 32;
        // fused-kernel shape: 32 / 64 / 128 queries per workgroup; 2 workgroups per CU in flight
        const bool shape8 = !ix->kn.shape4;   // default: ONE 8-wave workgroup per CU, 256 queries share each fetched row tile (diagnostics: two 4-wave workgroups of 128 queries)
        const int nqt = (shape8 && tiles > 4) ? 8 : tiles > 2 ? 4 : (int)tiles;
        const uint32_t n_super = (tiles + nqt - 1) / nqt;          // workgroups along the query axis (1 or 2)
        const float* qp0 = qp + (size_t)q0 * ld;

        vdb::DenseParams dp{ix->d_rows, ld, n, qp0, round_up(nb, 32), ix->d_alpha, ix->d_beta, d_rowmask, S,
                            ix->cur->w_dense.p, S};
        vdb::launch_dense_scores(dp, s);

        vdb::SelectParams sp{};
        sp.keys = ix->cur->w_dense.p; sp.stride = S; sp.counts = nullptr; sp.n_fixed = S; sp.cap = S; sp.kk = kp;
        sp.out_stride = kp;
        if (small) {
            sp.out_keys = ix->cur->w_cand.p; sp.out_cnt = d_cand_cnt; sp.out_thr = nullptr; sp.ovf = nullptr;
            vdb::launch_select(sp, nb, s);
        } else {
            // thresholds: the sample's kp-th score (padding queries were given -inf by query_prep)
            sp.out_keys = ix->cur->w_samp.p; sp.out_cnt = d_cnt_a; sp.out_thr = thr + q0; sp.ovf = nullptr;
            vdb::launch_select(sp, nb, s);
            const uint32_t n_wg = std::min<uint32_t>((nqt == 8 ? 1u : 2u) * (uint32_t)ix->n_cu / n_super, (n + 31) / 32);
            const uint32_t n_sub = vdb::fused_subpools_per_query(nqt, n_wg);
            vdb::FusedParams fp{ix->d_rows, ld, n, qp, q0, ix->d_alpha, ix->d_beta, d_rowmask ? d_rowmask : ix->d_live,
                                thr, ix->cur->w_pool.p - (size_t)q0 * n_sub * capl,
                                ix->cur->w_subcnt.p - (size_t)q0 * n_sub, capl, n_wg,
                                ix->kn.fused_ablate};
            const bool prof = ix->profile && !ix->cur->stats[8];       // with the screening tier on, ITS kernel is the one timed
            if (prof) HIP_TRY(hipEventRecord(ix->ev0, s));
            // 256-query passes: LDS-DMA staging, 3-image ring with the barrier in mid-stage; smaller batches: the
            // register-staged 128/64/32-query shapes.  (Diagnostics build: the 2-image and register-staged A/B variants.)
#ifdef VDB_DIAG
            if (nqt == 8 && ix->kn.regstage) vdb::launch_fused(fp, nqt, n_super, s);
            else if (nqt == 8 && ix->kn.dma2) vdb::launch_fused_dma(fp, n_super, s);
            else
#endif
            if (nqt == 8) vdb::launch_fused_dma3(fp, n_super, s);
            else vdb::launch_fused(fp, nqt, n_super, s);
            if (prof) {
                // one super-tile per event pair: wait here so the pair can be reused (profiling mode only)
                HIP_TRY(hipEventRecord(ix->ev1, s));
                HIP_TRY(hipEventSynchronize(ix->ev1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
                ix->cur->stats[7] += (uint64_t)((double)ms * 1e6);
            }
            ix->cur->stats[3] += n;
            vdb::SelectParams mp{};
            mp.keys = ix->cur->w_pool.p; mp.stride = 0; mp.counts = nullptr; mp.n_fixed = 0; mp.cap = 0;
            mp.sub_counts = ix->cur->w_subcnt.p; mp.n_sub = n_sub; mp.capl = capl;
            mp.kk = kp; mp.out_keys = ix->cur->w_cand.p; mp.out_stride = kp; mp.out_cnt = d_cand_cnt;
            mp.out_thr = nullptr; mp.ovf = d_ovf + q0; mp.summary = d_status + 1;
            vdb::launch_select(mp, nb, s);
        }
        vdb::RerankParams rp{};
        rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
        rp.qp = qp0; rp.qnorm = qnorm + q0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
        rp.rowmask = d_rowmask; rp.cand = ix->cur->w_cand.p; rp.cand_stride = kp; rp.cand_cnt = d_cand_cnt; rp.kp = kp;
        rp.metric = ix->metric; rp.k = (uint32_t)k; rp.eps_coef = eps; rp.nd2max_bits = ix->d_scalars;
        rp.out_ids = d_out_ids + (size_t)q0 * k; rp.out_dists = d_out_dists + (size_t)q0 * k;
        rp.out_counts = d_out_counts + q0; rp.out_stride = (uint32_t)k; rp.cert = d_cert + q0; rp.status = d_status;
        rp.thr = small ? nullptr : thr + q0;
        vdb::launch_rerank(rp, nb, s);
    }
    return VDB_OK;
}

// ------------------------------------------------------------------ tier: bf16 screening + certified re-rank
// Same structure, with the scores of the HBM-bound bf16 kernel (kernels_fused_bf16.hip): group minima of a row
// sample -> per-query threshold -> one pass over all rows keeping the keys under the threshold -> the kp smallest
// keys -> exact re-rank, certified with the bf16 error bound.  Queries come from ix->cur->w_qp / w_qb / w_qnorm.
int pass_bf16(Index* ix, hipStream_t s, uint32_t nq, size_t k, const Bf16Plan& pl, const uint32_t* d_rowmask,
              uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_cert, uint32_t* d_ovf,
              uint32_t* d_status, float* d_thr_next, bool allow_alt) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const uint32_t S = pl.S, kp = pl.kp, KT = pl.kt;
    const uint32_t M = vdb::fused_bf16_sample_groups(S);
    // private sub-pools of 256 slots: when the rows near a query are stored next to each other (data ordered by
    // cluster) most of the ~N*kt/S keys that pass land in ONE workgroup's four sub-pools; 4 x 256 slots hold about twice
    // the expected total, so that case stays on this tier instead of overflowing into the next.  The gather reads
    // counts and keys, never empty slots, so the capacity costs address space only (0.5 GB of workspace at 1M rows).
    const uint32_t capl = 256;
    const uint32_t n_wg = std::min<uint32_t>((uint32_t)ix->n_cu, (n + vdb::fused_bf16_tile_rows() - 1) / vdb::fused_bf16_tile_rows());
    // Batches above 256 queries: the WIDE filter kernel (kernels_fused_bf16w.hip, 128 rows x 512 queries per workgroup) serves
    // two 256-query blocks per fetch of the rows -- BASELINE config 3 (B = 1024) reads its shard twice per batch instead of
    // four times.  Not over the opt-in bf16 shadow rows (their kernel has the one shape); vdb_flat_set_wide(h, 0) turns it off.
    const bool use_wide = ix->wide && nq > SUPER && !shadow_usable(ix)
#ifdef VDB_DIAG
                          && ix->kn.fused_pipe
#endif
        ;
    const uint32_t n_wg_w = std::min<uint32_t>((uint32_t)ix->n_cu, (n + vdb::fused_bf16w_tile_rows() - 1) / vdb::fused_bf16w_tile_rows());
    const uint32_t n_sub = vdb::fused_bf16_subpools_per_query(std::max(n_wg, use_wide ? n_wg_w : 0u));   // (both kernels write the 4-sub-pool layout)
    const size_t pool_block = (size_t)SUPER * n_sub * capl, cnt_block = (size_t)SUPER * n_sub;
    // A batch above 256 queries takes several passes.  They are independent, so they ALTERNATE between
    // this context and the handle's other workspace and stream when that one is idle: the latency-bound tail of pass i (its
    // slowest re-rank workgroups, a few CUs) then runs beside the head of pass i+1 instead of in front of it.  The per-query
    // arrays (queries, thresholds, flags, outputs) are indexed by q0 and shared; only the pass-local buffers are doubled.
    Workspace* alt = nullptr;
    if (allow_alt && nq > (use_wide ? 2 * SUPER : SUPER) && !ix->profile && !ix->kn.rr_depth) {
        Workspace* o = (ix->cur == &ix->wsv[0]) ? &ix->wsv[1] : &ix->wsv[0];
        if (!o->busy) alt = o;
    }
    Workspace* const Wv[2] = {ix->cur, alt ? alt : ix->cur};
    const hipStream_t Sv[2] = {s, alt ? alt->stream : s};
    for (int t = 0; t < (alt ? 2 : 1); ++t) {
        Workspace* w = Wv[t];
        if ((rc = w->w_dense.ensure((size_t)SUPER * M))) return rc;
        if ((rc = w->w_cand.ensure((size_t)SUPER * kp))) return rc;
        if ((rc = w->w_samp.ensure((size_t)SUPER * std::max(kp, KT)))) return rc;
        if ((rc = w->w_pool.ensure((use_wide ? 2 : 1) * pool_block))) return rc;
        if ((rc = w->w_subcnt.ensure((use_wide ? 2 : 1) * cnt_block))) return rc;
        if ((rc = w->w_cnt.ensure(4 * SUPER + 16))) return rc;
    }
    // the sample pass runs over the compact bf16 copy of the sample rows when the row pitch allows (rebuilt here, before the
    // passes fork onto two streams, when rows were added since it was made -- mutators are refused while a search is in flight,
    // so nobody else is reading it; with ANOTHER search in flight and a different key this one keeps the f32 gather)
    bool sample_copy = ix->sample_cache && ld % 64 == 0 && !ix->kn.sample_block;
    if (sample_copy && (ix->sample16_n != n || ix->sample16_S != S)) {
        Workspace* o = (ix->cur == &ix->wsv[0]) ? &ix->wsv[1] : &ix->wsv[0];
        if (o->busy) sample_copy = false;
        else {
            const size_t need = (size_t)S * ld;
            if (ix->sample16_cap < need) {
                if (ix->d_sample16) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(ix->d_sample16); ix->d_sample16 = nullptr; ix->sample16_cap = 0; }
                if (hipMalloc((void**)&ix->d_sample16, need * 2) == hipSuccess) ix->sample16_cap = need;
                else { (void)hipGetLastError(); ix->d_sample16 = nullptr; sample_copy = false; }   // no memory for the optional copy: f32 gather
            }
            if (sample_copy) {
                vdb::launch_sample_to_bf16(ix->d_rows, ld, n, S, pl.shift, ix->d_sample16, s);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(s));                    // once per change of the rows: later searches on OTHER streams read it
                ix->sample16_n = n; ix->sample16_S = S;
            }
        }
    }
    if (alt) {                                                   // the other stream starts behind query_prep and the row mask
        if (!ix->ev_pass[0]) {
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_pass[0], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_pass[1], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(ix->ev_pass[0], s));
        HIP_TRY(hipStreamWaitEvent(Sv[1], ix->ev_pass[0], 0));
    }
    ix->cur->stats[4] = S;
    const float eps = eps_coef(ix);
    // thresholds: sample pass + threshold select per 256-query block.  A batch of several passes computes ALL of them first, on the
    // caller's stream: left to their own pass, the second stream's small kernels queue up behind the first pass's filter kernel
    // (which owns every CU for a millisecond) and the second filter pass starts late.
    auto thresholds_of = [&](Workspace* W, hipStream_t st, uint32_t qb0) {
        const uint32_t nb = std::min(SUPER, nq - qb0);
        vdb::FusedBf16Params fp{};
        fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n;
        fp.alpha = ix->d_alpha; fp.beta = ix->d_beta; fp.rowmask = d_rowmask ? d_rowmask : ix->d_live;
        fp.margin = ix->d_margin;
        fp.pool = W->w_pool.p; fp.pool_cnt = W->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
        fp.scalars = ix->d_scalars; fp.qmax_bits = d_status + 2;
        fp.ablate = ix->kn.bf16_ablate;
        fp.n_sample = S; fp.sample_shift = pl.shift;
        fp.sample_block = ix->kn.sample_block ? (n / (S / 256u)) : 0u; fp.minkeys = W->w_dense.p; fp.minkey_stride = M;
        fp.qb = ix->cur->w_qb.p + (size_t)qb0 * ld; fp.qg = ix->d_margin ? ix->cur->w_qg.p + qb0 : nullptr; fp.thr = ix->cur->w_thr.p + qb0;
        if (sample_copy) {
            vdb::FusedBf16Params sp16 = fp;
            sp16.rows16 = ix->d_sample16;
            vdb::launch_sample_s16(sp16, st);
        } else vdb::launch_sample_bf16(fp, (uint32_t)ix->n_cu, st);
        vdb::SelectParams sp{};
        sp.keys = W->w_dense.p; sp.stride = M; sp.counts = nullptr; sp.n_fixed = M; sp.cap = M; sp.kk = KT;
        sp.out_stride = KT; sp.out_keys = W->w_samp.p; sp.out_cnt = W->w_cnt.p; sp.out_thr = ix->cur->w_thr.p + qb0; sp.ovf = nullptr;
        if (ix->d_margin) { sp.shift_g = ix->cur->w_qg.p + qb0; sp.shift_m_bits = ix->d_scalars + 4; }   // plain-score sample -> lower-bound units
        vdb::launch_thr_select(sp, nb, st);
    };
    // (measured at config 3's shape: four 256-query passes 3.26 -> 3.09 ms with the thresholds first; the two 512-query passes
    // are better off with their own -- 2.75 against 2.84 ms: the second pass's thresholds then run beside the first pass's tail)
    const uint32_t n_passes = use_wide ? (nq + 2 * SUPER - 1) / (2 * SUPER) : (nq + SUPER - 1) / SUPER;
    const bool thr_first = alt != nullptr && n_passes > 2;
    if (thr_first) {
        for (uint32_t qb0 = 0; qb0 < nq; qb0 += SUPER) thresholds_of(ix->cur, s, qb0);
        HIP_TRY(hipEventRecord(ix->ev_pass[0], s));               // (re-recorded: the other stream starts behind the thresholds)
        HIP_TRY(hipStreamWaitEvent(Sv[1], ix->ev_pass[0], 0));
    }
    for (uint32_t q0 = 0, pass = 0; q0 < nq; ++pass) {
        const bool wide = use_wide && nq - q0 > SUPER;            // two 256-query blocks share this pass's fetch of the rows
        const uint32_t n_blocks = wide ? 2u : 1u;
        Workspace* const W = Wv[pass & 1];                        // pass-local buffers
        const hipStream_t s = Sv[pass & 1];                       // (shadows the caller's stream inside the loop)
        uint32_t* d_cand_cnt = W->w_cnt.p + 2 * SUPER;
        vdb::FusedBf16Params fp{};
        fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n;
        fp.alpha = ix->d_alpha; fp.beta = ix->d_beta; fp.rowmask = d_rowmask ? d_rowmask : ix->d_live;
        fp.margin = ix->d_margin;
        fp.pool = W->w_pool.p; fp.pool_cnt = W->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
        fp.scalars = ix->d_scalars; fp.qmax_bits = d_status + 2;
        fp.ablate = ix->kn.bf16_ablate;
        fp.n_sample = S; fp.sample_shift = pl.shift;
        fp.sample_block = ix->kn.sample_block ? (n / (S / 256u)) : 0u; fp.minkeys = W->w_dense.p; fp.minkey_stride = M;
        // ---- thresholds of the pass's blocks (unless they were all computed up front)
        if (!thr_first)
            for (uint32_t b = 0; b < n_blocks; ++b) thresholds_of(W, s, q0 + b * SUPER);
        // ---- ONE pass over the rows for all of them
        fp.qb = ix->cur->w_qb.p + (size_t)q0 * ld; fp.qg = ix->d_margin ? ix->cur->w_qg.p + q0 : nullptr; fp.thr = ix->cur->w_thr.p + q0;
        if (ix->profile) HIP_TRY(hipEventRecord(ix->ev0, s));
        if (wide) {
            fp.n_wg = n_wg_w; fp.pool_block_stride = pool_block; fp.cnt_block_stride = cnt_block;
            vdb::launch_fused_bf16w(fp, s);
        } else launch_filter_pass(ix, fp, s);
        if (ix->profile) {
            HIP_TRY(hipEventRecord(ix->ev1, s));
            HIP_TRY(hipEventSynchronize(ix->ev1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            ix->cur->stats[7] += (uint64_t)((double)ms * 1e6);
        }
#ifdef VDB_DIAG
        // An ablated launch leaves wrong pools behind; if the step went on with them every query would fall through to the
        // slower tiers, and the extra milliseconds of f32 MFMA work change the clock the NEXT timed launch runs at (ablation
        // arms with broken results read 10-25 us low for that reason alone).  So the ablated launch is the timed one, and an
        // unablated launch (untimed) overwrites its pools: every arm of an A/B then runs the same step around the kernel.
        if (fp.ablate) {
            fp.ablate = 0;
            if (wide) vdb::launch_fused_bf16w(fp, s); else launch_filter_pass(ix, fp, s);
            ix->cur->stats[3] += n;
        }
#endif
        ix->cur->stats[3] += n;

        // ---- per block: the smallest pooled keys, then the exact re-rank
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const uint32_t qb0 = q0 + b * SUPER, nb = std::min(SUPER, nq - qb0);
            vdb::SelectParams mp{};
            mp.keys = W->w_pool.p + (size_t)b * pool_block; mp.stride = 0; mp.counts = nullptr; mp.n_fixed = 0; mp.cap = 0;
            mp.sub_counts = W->w_subcnt.p + (size_t)b * cnt_block; mp.n_sub = vdb::fused_bf16_subpools_per_query(wide ? n_wg_w : n_wg);
            mp.capl = capl; mp.wg_major = 1;
            mp.kk = kp; mp.out_keys = W->w_cand.p; mp.out_stride = kp; mp.out_cnt = d_cand_cnt;
            mp.out_thr = nullptr; mp.ovf = d_ovf + qb0; mp.summary = d_status + 1;
            vdb::launch_select(mp, nb, s);

            vdb::RerankParams rp{};
            rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
            rp.qp = ix->cur->w_qp.p + (size_t)qb0 * ld; rp.qnorm = ix->cur->w_qnorm.p + qb0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
            rp.rowmask = d_rowmask; rp.cand = W->w_cand.p; rp.cand_stride = kp; rp.cand_cnt = d_cand_cnt; rp.kp = kp;
            rp.metric = ix->metric; rp.k = (uint32_t)k; rp.eps_coef = eps; rp.nd2max_bits = ix->d_scalars;
            rp.out_ids = d_out_ids + (size_t)qb0 * k; rp.out_dists = d_out_dists + (size_t)qb0 * k;
            rp.out_counts = d_out_counts + qb0; rp.out_stride = (uint32_t)k; rp.cert = d_cert + qb0; rp.status = d_status;
            rp.thr = ix->cur->w_thr.p + qb0;
            rp.qerr = ix->cur->w_qerr.p + qb0; rp.c_acc = c_acc_bf16(ix); rp.lb_scores = ix->d_margin ? 1u : 0u;
            rp.kp_first = round_up((uint32_t)k + 38u, 16u); rp.kp_step = 32;
            rp.thr_next = d_thr_next ? d_thr_next + qb0 : nullptr;
            // diagnostics build: the first re-rank round overridden, the depth each query ended at printed
            if (ix->kn.kp_first) rp.kp_first = ix->kn.kp_first;
            const bool dump_depth = ix->kn.rr_depth;
            if (dump_depth) {
                if ((rc = W->w_depth.ensure(SUPER * 17))) return rc;      // depth[q], then 8 x 64-bit phase stamps per query
                rp.depth = W->w_depth.p;
            }
            vdb::launch_rerank(rp, nb, s);
            if (dump_depth) {
                std::vector<uint32_t> dep((size_t)SUPER * 17);
                HIP_TRY(hipMemcpyAsync(dep.data(), W->w_depth.p, dep.size() * 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                // phase stamps (s_memrealtime, 100 MHz): 0 start, 1 query row in LDS, 2 round 1 staged+folded, 3 sorted, 4 depth decided, 5 last round folded, 6 end
                const uint64_t* st64 = reinterpret_cast<const uint64_t*>(dep.data() + SUPER);
                uint64_t t0 = ~0ull;
                for (uint32_t q = 0; q < nb; ++q) t0 = std::min(t0, st64[(size_t)q * 8]);
                for (int ph = 0; ph < 7; ++ph) {
                    std::vector<double> v(nb);
                    for (uint32_t q = 0; q < nb; ++q) v[q] = (double)(st64[(size_t)q * 8 + ph] - t0) * 0.01;
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "[vdb] re-rank phase %d at us: min %.2f median %.2f p90 %.2f max %.2f\n", ph, v[0], v[nb / 2], v[(size_t)nb * 9 / 10], v[nb - 1]);
                }
                std::sort(dep.begin(), dep.begin() + nb);
                fprintf(stderr, "[vdb] re-rank depth of %u queries: min %u  p25 %u  median %u  p75 %u  p95 %u  max %u  (first round %u)\n", nb,
                        dep[0], dep[nb / 4], dep[nb / 2], dep[(size_t)nb * 3 / 4], dep[(size_t)nb * 95 / 100], dep[nb - 1], rp.kp_first);
            }
        }
        q0 += n_blocks * SUPER;
    }
    if (alt) {                                                   // the caller's stream continues behind BOTH chains
        HIP_TRY(hipEventRecord(ix->ev_pass[1], Sv[1]));
        HIP_TRY(hipStreamWaitEvent(s, ix->ev_pass[1], 0));
    }
    return VDB_OK;
}

// ------------------------------------------------------------------ the batched search
// ------------------------------------------------------------------ tier 0b: the re-threshold pass
// For queries the screening tier re-ranked to its depth limit without a certificate, the k-th exact distance found so
// far still bounds the answer: rerank_kernel turned it into a score cut above which no row can enter the top k.  The
// queries are gathered into a compact block, the HBM-bound filter pass runs once more with those cuts as thresholds,
// and EVERY key that passes (up to 2048 per query) is re-ranked exactly.  Exact by construction; a query whose list does
// not fit (pool overflow, more than 2048 keys) keeps its flag and goes on to the next tier.
// todo: batch indices; cuts: their score cuts.  On return flags2 (host) holds cert / overflow per compact query.
int pass_rethreshold(Index* ix, hipStream_t s, const std::vector<uint32_t>& todo, const std::vector<float>& cuts, size_t k,
                     const uint32_t* d_rowmask, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                     uint32_t* d_status, std::vector<uint32_t>& flags2) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const uint32_t nf = (uint32_t)todo.size(), nfp = round_up(nf, SUPER);
    constexpr uint32_t KMAX = 2048;                              // keys re-ranked per query at most (select capacity)
    const uint32_t capl = 256;
    const uint32_t n_wg = std::min<uint32_t>((uint32_t)ix->n_cu, (n + vdb::fused_bf16_tile_rows() - 1) / vdb::fused_bf16_tile_rows());
    const uint32_t n_sub = vdb::fused_bf16_subpools_per_query(n_wg);
    if ((rc = ix->cur->w2_qp.ensure((size_t)nfp * ld))) return rc;
    if ((rc = ix->cur->w2_qnorm.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_thr.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qerr.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qg.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qb.ensure((size_t)nfp * ld))) return rc;
    if ((rc = ix->cur->w2_outi.ensure((size_t)nf * k))) return rc;
    if ((rc = ix->cur->w2_outd.ensure((size_t)nf * k))) return rc;
    if ((rc = ix->cur->w2_outc.ensure(nf))) return rc;
    if ((rc = ix->cur->w2_flags.ensure(2 * (size_t)nf))) return rc;
    if ((rc = ix->cur->w2_qidx.ensure(nf))) return rc;
    if ((rc = ix->cur->w2_cand.ensure((size_t)SUPER * KMAX))) return rc;
    if ((rc = ix->cur->w_pool.ensure((size_t)SUPER * n_sub * capl))) return rc;
    if ((rc = ix->cur->w_subcnt.ensure((size_t)SUPER * n_sub))) return rc;
    uint32_t* d_cert2 = ix->cur->w2_flags.p;
    uint32_t* d_ovf2 = ix->cur->w2_flags.p + nf;
    HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, todo.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
    vdb::launch_gather_queries(ix->cur->w_qp.p, ix->cur->w_qnorm.p, ld, ix->cur->w2_qidx.p, nf, nfp, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, s);
    // bf16 image, |q - bf16(q)| and zeroed flags of the compact block (the rows are already padded: dim = ld)
    vdb::QueryPrepParams qp{ix->cur->w2_qp.p, ld, nf, ix->cur->w2_qp.p, ld, nfp, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, vdb::EUCLID, d_status,
                            ix->cur->w2_qb.p, ix->cur->w2_qerr.p, ix->d_margin ? ix->cur->w2_qg.p : nullptr, margin_plan(ix).kappa, d_cert2, d_ovf2};
    vdb::launch_query_prep(qp, s);
    HIP_TRY(hipMemcpyAsync(ix->cur->w2_thr.p, cuts.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));   // padding queries keep -inf
    uint32_t* d_cand_cnt = ix->cur->w_cnt.p + 2 * SUPER;
    for (uint32_t q0 = 0; q0 < nf; q0 += SUPER) {
        const uint32_t nb = std::min(SUPER, nf - q0);
        vdb::FusedBf16Params fp{};
        fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n; fp.qb = ix->cur->w2_qb.p + (size_t)q0 * ld;
        fp.alpha = ix->d_alpha; fp.beta = ix->d_beta; fp.rowmask = d_rowmask ? d_rowmask : ix->d_live;
        fp.margin = ix->d_margin; fp.qg = ix->d_margin ? ix->cur->w2_qg.p + q0 : nullptr;
        fp.thr = ix->cur->w2_thr.p + q0; fp.pool = ix->cur->w_pool.p; fp.pool_cnt = ix->cur->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
        fp.scalars = ix->d_scalars; fp.qmax_bits = d_status + 2;
        launch_filter_pass(ix, fp, s);
        ix->cur->stats[3] += n;
        vdb::SelectParams mp{};
        mp.keys = ix->cur->w_pool.p; mp.sub_counts = ix->cur->w_subcnt.p; mp.n_sub = n_sub; mp.capl = capl; mp.wg_major = 1;
        mp.kk = KMAX; mp.out_keys = ix->cur->w2_cand.p; mp.out_stride = KMAX; mp.out_cnt = d_cand_cnt;
        mp.ovf = d_ovf2 + q0; mp.summary = nullptr; mp.flag_truncation = 1;
        vdb::launch_select(mp, nb, s);
        vdb::RerankParams rp{};
        rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
        rp.qp = ix->cur->w2_qp.p + (size_t)q0 * ld; rp.qnorm = ix->cur->w2_qnorm.p + q0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
        rp.rowmask = d_rowmask; rp.cand = ix->cur->w2_cand.p; rp.cand_stride = KMAX; rp.cand_cnt = d_cand_cnt; rp.kp = KMAX;
        rp.metric = ix->metric; rp.k = (uint32_t)k; rp.nd2max_bits = ix->d_scalars;
        rp.out_ids = ix->cur->w2_outi.p + (size_t)q0 * k; rp.out_dists = ix->cur->w2_outd.p + (size_t)q0 * k;
        rp.out_counts = ix->cur->w2_outc.p + q0; rp.out_stride = (uint32_t)k; rp.cert = d_cert2 + q0; rp.status = d_status;
        vdb::launch_rerank_all(rp, nb, s);
    }
    HIP_TRY(hipGetLastError());
    flags2.assign(2 * (size_t)nf, 0u);
    HIP_TRY(hipMemcpyAsync(flags2.data(), ix->cur->w2_flags.p, 2 * (size_t)nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // only the queries this pass answered completely are written back
    std::vector<uint32_t> good;
    for (uint32_t j = 0; j < nf; ++j) if (flags2[j] && !flags2[nf + j]) good.push_back(j);
    if (!good.empty()) {
        // scatter compact results j -> batch position todo[j] (the scatter kernel walks a (source, destination) list)
        std::vector<uint32_t> src_dst(2 * good.size());
        for (size_t i = 0; i < good.size(); ++i) { src_dst[i] = good[i]; src_dst[good.size() + i] = todo[good[i]]; }
        if ((rc = ix->cur->w2_qidx.ensure(2 * good.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, src_dst.data(), src_dst.size() * 4, hipMemcpyHostToDevice, s));
        vdb::launch_scatter_results_list(ix->cur->w2_outi.p, ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_qidx.p, ix->cur->w2_qidx.p + good.size(),
                                         (uint32_t)good.size(), (uint32_t)k, d_out_ids, d_out_dists, d_out_counts, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
    }
    return VDB_OK;
}

// ------------------------------------------------------------------ the direct path of small indexes
// Index::search as it stands (flat_index.rs:52-65) -- every row's exact distance, the k smallest by (distance, id) -- for an
// index of at most 16384 rows and a batch of at most DIRECT_MAX_Q queries: BASELINE configs[0] (benches/search_bench.rs:18-33:
// 10k x 128, ONE query).  At that size the tiered pipeline is five launches of latency (query preparation, dense MFMA scores,
// select, re-rank, flag copy: ~90 us per call); here it is TWO kernels and one stream synchronisation -- small_scan_kernel reads
// the raw queries (from wherever they are: the host-pointer entry point hands over MAPPED host memory, so no copy is enqueued at
// all) and writes one exact key per row, select_kernel in EMIT mode writes ids, distances, counts and the status word straight
// into the caller's buffers (mapped host memory again for the host-pointer entry point).  Exact by construction: no certificate.
bool direct_eligible(const Index* ix, size_t n_rows, size_t nq, size_t k) {
    return !(ix->tiers & VDB_TIERS_NO_DIRECT) && n_rows > 0 && n_rows <= SMALL_N && nq > 0 && nq <= DIRECT_MAX_Q && k > 0 && k <= MAX_SELECT;
}

int ensure_host_io(Index* ix, size_t bytes) {
    Workspace* W = ix->cur;
    if (bytes <= W->h_io_bytes) return VDB_OK;
    if (W->h_io) (void)hipHostFree(W->h_io);
    W->h_io = W->d_h_io = nullptr; W->h_io_bytes = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 2, 1u << 16);
    HIP_TRY(hipHostMalloc((void**)&W->h_io, cap, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&W->d_h_io, W->h_io, 0));
    W->h_io_bytes = cap;
    return VDB_OK;
}

static int search_direct(Index* ix, hipStream_t s, const float* d_q, uint32_t nq, size_t k, const uint32_t* d_rowmask,
                         uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts) {
    int rc;
    Workspace* W = ix->cur;
    const uint32_t n = ix->n_uploaded;
    if ((rc = ensure_ranks(ix))) return rc;
    // every workgroup of the scan keeps its k smallest keys only (k < 256): the select ranks n/256 * k keys instead of n
    const uint32_t keep = k < 256 ? (uint32_t)k : 0u;
    const uint32_t n_keys = keep ? vdb::small_scan_groups(n) * keep : n;
    if ((rc = W->w_exact.ensure((size_t)nq * std::max(n_keys, n)))) return rc;
    if ((rc = W->w_exsel.ensure((size_t)nq * MAX_SELECT + 8))) return rc;
    if ((rc = W->w_cnt.ensure(4 * SUPER + 16))) return rc;
    if (!W->dstat_ready) {
        if ((rc = W->w_dstat.ensure(4))) return rc;
        if (!W->h_dstat) {
            HIP_TRY(hipHostMalloc((void**)&W->h_dstat, 16 * sizeof(uint32_t), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&W->d_h_dstat, W->h_dstat, 0));
        }
        HIP_TRY(hipMemsetAsync(W->w_dstat.p, 0, 16, s));
        W->dstat_ready = true;
    }
    vdb::SmallScanParams sp{ix->d_rows, ix->ld, ix->dim, n, d_q, nq, ix->d_nd, d_rowmask, ix->ids_monotone ? nullptr : ix->d_idrank.p,
                            ix->metric, W->w_exact.p, n_keys, keep, W->w_dstat.p};
    vdb::launch_small_scan(sp, s);
    vdb::SelectParams mp{};
    mp.keys = W->w_exact.p; mp.stride = n_keys; mp.counts = nullptr; mp.n_fixed = n_keys; mp.cap = n_keys;
    mp.kk = (uint32_t)k; mp.out_keys = W->w_exsel.p; mp.out_stride = MAX_SELECT; mp.out_cnt = W->w_cnt.p;
    mp.emit_ids = d_out_ids; mp.emit_dists = d_out_dists; mp.emit_counts = d_out_counts; mp.emit_stride = (uint32_t)k;
    mp.emit_rank2row = ix->ids_monotone ? nullptr : ix->d_rank2row.p; mp.emit_row_ids = ix->d_row_ids;
    mp.emit_status_in = W->w_dstat.p; mp.emit_status = W->d_h_dstat;
    vdb::launch_select(mp, nq, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    uint32_t status = 0;
    for (uint32_t q = 0; q < nq; ++q) status |= W->h_dstat[q];
    ix->cur->stats[1] = nq;                                         // answered by an exact scan
    if (status) {                                                  // (rare: leave the device word clean for the next search)
        HIP_TRY(hipMemsetAsync(W->w_dstat.p, 0, 16, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (status & vdb::ST_ZERO_QUERY)
        return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    if (status & vdb::ST_NAN) return fail(VDB_ERR_NAN, "NaN distance (the reference panics here, flat_index.rs:62)");
    return VDB_OK;
}

// Part 1: checks, workspace, and the FIRST tier enqueued on the stream -- no host synchronisation unless the search is
// one of the cases answered completely here (empty store, k = 0, k too large for the MFMA tiers).
int search_part1(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                 size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                 hipStream_t user_stream, bool allow_alt) {
    int rc;
    ix->cur->ctx.pending = false;
    if ((rc = set_device(ix))) return rc;
    if ((rc = flush(ix))) return rc;
    if (nq == 0) return VDB_OK;
    // all launches of this search go to the caller's stream when one is given (so that the caller's
    // events bracket them); the workspace is protected by the handle mutex and the final sync
    hipStream_t s = user_stream ? user_stream : ix->cur->stream;
    memset(ix->cur->stats, 0, sizeof(ix->cur->stats));
    ix->cur->stats[14] = shadow_usable(ix) ? 1u : 0u;   // the screening pass reads the bf16 shadow rows
    ix->cur->stats[15] = ix->kn.any ? 1u : 0u;          // diagnostics build with a knob set: the run is NOT covered by the exactness guarantee
    const auto t_entry = std::chrono::steady_clock::now();
    auto since = [&]() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(); };
    size_t total_rows = ix->n_live + ix->misfits.size();
    if (total_rows == 0 || k == 0) {   // storage.rs:218-220: empty store -> Ok(vec![]) before any check
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        return VDB_OK;
    }
    // distance.rs:21-26: the first row whose dimension differs from the query's fails the search
    if (ix->n_live && ix->dim != dim) return fail_dim(dim, ix->dim);
    for (auto& kv : ix->misfits)
        if (kv.second.size() != dim) return fail_dim(dim, kv.second.size());
    if (!ix->misfits.empty()) {
        // every stored row has the query's dimension but none is on the device (zero-length rows)
        return fail(VDB_ERR_INVALID_ARGUMENT, "zero-dimensional vectors are not searchable");
    }
    if (ix->metric == vdb::COSINE) {
        if ((rc = ensure_zero_count(ix))) return rc;
        if (ix->zero_live)   // distance.rs:51-55 aborts the whole search (flat_index.rs:57-60)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    }
    if (nq > 0x7fffffffull / 2 || k > 0x7fffffffull) return fail(VDB_ERR_INVALID_ARGUMENT, "batch too large");
    if (ix->dim > 16384) return fail(VDB_ERR_INVALID_ARGUMENT, "dimension %u exceeds the supported 16384", ix->dim);

    const uint32_t n = ix->n_uploaded;
    const uint32_t ld = ix->ld;
    const uint32_t nq32 = (uint32_t)nq;
    const uint32_t bp_all = round_up(nq32, SUPER);
    const uint32_t kp = pick_kp(k);

    // ---- small index, a few queries: the direct exact path (two kernels, answered completely here)
    if (direct_eligible(ix, n, nq, k)) {
        const uint32_t* d_rowmask = (ix->n_live == n) ? nullptr : ix->d_live;
        if (d_idmask) {
            if ((rc = ix->cur->w_rowmask.ensure((n + 31) / 32))) return rc;
            vdb::launch_build_rowmask(ix->d_row_ids, d_rowmask, d_idmask, mask_bits, n, ix->cur->w_rowmask.p, s);
            d_rowmask = ix->cur->w_rowmask.p;
        }
        rc = search_direct(ix, s, d_q, nq32, k, d_rowmask, d_out_ids, d_out_dists, d_out_counts);
        ix->cur->stats[10] = ix->cur->stats[11] = ix->cur->stats[12] = since();
        return rc;
    }

    // ---- workspace
    if ((rc = ix->cur->w_qp.ensure((size_t)bp_all * ld))) return rc;
    if ((rc = ix->cur->w_qnorm.ensure(bp_all))) return rc;
    if ((rc = ix->cur->w_thr.ensure(bp_all))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4 + 3 * (size_t)nq32))) return rc;      // status block | cert | overflow | score cut per query
    if (ix->cur->h_flags_n < 4 + 3 * (size_t)nq32) {
        if (ix->cur->h_flags) (void)hipHostFree(ix->cur->h_flags);
        ix->cur->h_flags = nullptr;
        ix->cur->h_flags_n = 0;
        size_t want = 4 + 3 * (size_t)nq32 + 1024;
        HIP_TRY(hipHostMalloc((void**)&ix->cur->h_flags, want * 4, hipHostMallocDefault));
        ix->cur->h_flags_n = want;
    }
    uint32_t* d_status = ix->cur->w_flags.p;        // [0] status bits
    uint32_t* d_cert = ix->cur->w_flags.p + 4;      // [nq]
    uint32_t* d_ovf = d_cert + nq32;           // [nq]
    // the per-query flags are zeroed by query_prep; the 16-byte status block only needs a memset when the last
    // search left it set (or the buffer is new) -- one launch less at the head of every search
    const bool flags_by_prep = kp != 0 || (ix->screen && plan_bf16(ix, n, k).kp);
    if (!flags_by_prep) HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, (4 + 3 * (size_t)nq32) * 4, s));
    else if (ix->cur->status_dirty || ix->cur->w_flags.p != ix->cur->status_buf) {
        HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, 16, s));
        ix->cur->status_buf = ix->cur->w_flags.p;
    }
    ix->cur->status_dirty = true;                               // until a clean status word has been read back

    // ---- eligibility mask: tombstones, optionally AND the caller's id filter
    const uint32_t* d_rowmask = (ix->n_live == n) ? nullptr : ix->d_live;
    if (d_idmask) {
        if ((rc = ix->cur->w_rowmask.ensure((n + 31) / 32))) return rc;
        vdb::launch_build_rowmask(ix->d_row_ids, d_rowmask, d_idmask, mask_bits, n, ix->cur->w_rowmask.p, s);
        d_rowmask = ix->cur->w_rowmask.p;
    }

    // ---- queries: zero-padded copy + exact-order norms
    {
        uint16_t* qb = nullptr;
        if (ix->screen && plan_bf16(ix, n, k).kp) {
            if ((rc = ix->cur->w_qb.ensure((size_t)bp_all * ld))) return rc;
            if ((rc = ix->cur->w_qerr.ensure(bp_all))) return rc;
            if ((rc = ix->cur->w_qg.ensure(bp_all))) return rc;
            qb = ix->cur->w_qb.p;
        }
        vdb::QueryPrepParams qp{d_q, (uint32_t)dim, nq32, ix->cur->w_qp.p, ld, bp_all, ix->cur->w_qnorm.p, ix->cur->w_thr.p, ix->metric, d_status, qb,
                                ix->cur->w_qerr.p, (qb && ix->d_margin) ? ix->cur->w_qg.p : nullptr, margin_plan(ix).kappa,
                                flags_by_prep ? d_cert : nullptr, flags_by_prep ? d_ovf : nullptr};
        vdb::launch_query_prep(qp, s);
    }

    if (kp == 0 && !(ix->screen && plan_bf16(ix, n, k).kp)) {
        // large k: exact scan for every query
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ix->cur->h_flags[0] & vdb::ST_ZERO_QUERY)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
        for (uint32_t q = 0; q < nq32; ++q) {
            if ((rc = exact_one(ix, s, q, k, d_rowmask, d_out_ids + (size_t)q * k, d_out_dists + (size_t)q * k,
                                d_out_counts + q)))
                return rc;
        }
        ix->cur->stats[1] = nq32;
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ix->cur->h_flags[0] & vdb::ST_NAN) return fail(VDB_ERR_NAN, "NaN distance (the reference panics here, flat_index.rs:62)");
        return VDB_OK;
    }

    // ---- tiers.  Large indexes: the bf16 screening tier first (HBM-bound pass), the queries it cannot certify
    // are re-run as a compact block by the f32 MFMA tier; whatever that cannot certify goes to the exact scan.
    const Bf16Plan pl16 = ix->screen ? plan_bf16(ix, n, k) : Bf16Plan{};
    const uint32_t kp16 = pl16.kp;
    if ((rc = ix->cur->w_cnt.ensure(4 * SUPER + 16))) return rc;
    if (kp16) {
        ix->cur->stats[8] = 1;
        ix->cur->stats[5] = kp16;
        if ((rc = pass_bf16(ix, s, nq32, k, pl16, d_rowmask, d_out_ids, d_out_dists, d_out_counts, d_cert, d_ovf, d_status,
                            reinterpret_cast<float*>(d_ovf + nq32), allow_alt)))
            return rc;
    } else {
        if ((rc = pass_f32(ix, s, ix->cur->w_qp.p, ix->cur->w_qnorm.p, ix->cur->w_thr.p, nq32, k, kp, d_rowmask, d_out_ids, d_out_dists,
                           d_out_counts, d_cert, d_ovf, d_status)))
            return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, (4 + 3 * (size_t)nq32) * 4, hipMemcpyDeviceToHost, s));
    ix->cur->stats[10] = since();                       // host time until everything of the first tier is enqueued, ns
    Workspace::SearchCtx& c = ix->cur->ctx;
    c.pending = true; c.nq32 = nq32; c.kp = kp; c.kp16 = kp16; c.k = k; c.s = s; c.d_rowmask = d_rowmask;
    c.d_out_ids = d_out_ids; c.d_out_dists = d_out_dists; c.d_out_counts = d_out_counts; c.t_entry = t_entry;
    return VDB_OK;
}

// Part 2: wait for the first tier, read its flags, run the fallback tiers for the queries it could not certify.
// *changed (may be null) tells whether outputs were rewritten after part 1's pass.
int search_part2(Index* ix, int* changed) {
    if (changed) *changed = 0;
    Workspace::SearchCtx& c = ix->cur->ctx;
    if (!c.pending) return VDB_OK;
    c.pending = false;
    int rc;
    const uint32_t nq32 = c.nq32, kp = c.kp, kp16 = c.kp16;
    const size_t k = c.k;
    hipStream_t s = c.s;
    const uint32_t* d_rowmask = c.d_rowmask;
    uint64_t* d_out_ids = c.d_out_ids; float* d_out_dists = c.d_out_dists; uint32_t* d_out_counts = c.d_out_counts;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    uint32_t* d_status = ix->cur->w_flags.p;
    const auto t_entry = c.t_entry;
    auto since = [&]() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(); };
    HIP_TRY(hipStreamSynchronize(s));
    ix->cur->stats[11] = since();                       // ... until the first tier's flags are on the host, ns
    uint32_t status = ix->cur->h_flags[0];
    if (status & vdb::ST_ZERO_QUERY)
        return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    // vdb_flat_set_tiers: forced hand-over to the slower tiers (tests); every tier returns the same results
    const bool force_exact = (ix->tiers & VDB_TIERS_FORCE_EXACT) != 0;
    const bool force_f32 = (ix->tiers & VDB_TIERS_FORCE_F32) != 0;
    const bool no_rethr = (ix->tiers & VDB_TIERS_NO_RETHRESHOLD) != 0;
    std::vector<uint32_t> todo;
    for (uint32_t q = 0; q < nq32; ++q) {
        bool cert = ix->cur->h_flags[4 + q] != 0, ovf = ix->cur->h_flags[4 + nq32 + q] != 0;
        if (ovf) ++ix->cur->stats[2];
        if (!cert) ++ix->cur->stats[6];
        if (cert && !ovf && !force_exact && !(kp16 && force_f32)) continue;
        todo.push_back(q);
    }
    if (changed && !todo.empty()) *changed = 1;
    if (kp16 && !todo.empty() && !no_rethr && !force_exact && !force_f32) {
        // ---- tier 0b: queries with a known score cut get one more HBM-bound pass with that cut as the threshold
        const uint32_t* h_ovf = ix->cur->h_flags + 4 + nq32;
        const float* h_cut = reinterpret_cast<const float*>(ix->cur->h_flags + 4 + 2 * (size_t)nq32);
        std::vector<uint32_t> sel, rest;
        std::vector<float> cuts;
        for (uint32_t q : todo) {
            const float c = h_cut[q];
            if (!h_ovf[q] && c == c && std::isfinite(c)) { sel.push_back(q); cuts.push_back(c); }
            else rest.push_back(q);
        }
        if (!sel.empty()) {
            std::vector<uint32_t> fl;
            if ((rc = pass_rethreshold(ix, s, sel, cuts, k, d_rowmask, d_out_ids, d_out_dists, d_out_counts, d_status, fl))) return rc;
            const uint32_t nf = (uint32_t)sel.size();
            for (uint32_t j = 0; j < nf; ++j) {
                if (fl[j] && !fl[nf + j]) ++ix->cur->stats[13];
                else { rest.push_back(sel[j]); if (fl[nf + j]) ++ix->cur->stats[2]; }
            }
            std::sort(rest.begin(), rest.end());
        }
        todo.swap(rest);
    }
    if (kp16 && !todo.empty()) {
        // ---- second tier: the uncertified queries as one compact block through the f32 MFMA pipeline
        const uint32_t nf = (uint32_t)todo.size(), nfp = round_up(nf, SUPER);
        ix->cur->stats[9] = nf;
        if ((rc = ix->cur->w2_qp.ensure((size_t)nfp * ld))) return rc;
        if ((rc = ix->cur->w2_qnorm.ensure(nfp))) return rc;
        if ((rc = ix->cur->w2_thr.ensure(nfp))) return rc;
        if ((rc = ix->cur->w2_outi.ensure((size_t)nf * k))) return rc;
        if ((rc = ix->cur->w2_outd.ensure((size_t)nf * k))) return rc;
        if ((rc = ix->cur->w2_outc.ensure(nf))) return rc;
        if ((rc = ix->cur->w2_flags.ensure(2 * (size_t)nf))) return rc;
        if ((rc = ix->cur->w2_qidx.ensure(nf))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, todo.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(ix->cur->w2_flags.p, 0, 2 * (size_t)nf * 4, s));
        vdb::launch_gather_queries(ix->cur->w_qp.p, ix->cur->w_qnorm.p, ld, ix->cur->w2_qidx.p, nf, nfp, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p,
                                   ix->cur->w2_thr.p, s);
        if (kp == 0) {
            // k too large for the f32 tier as well: straight to the exact scan (flags stay 0 = uncertified)
        } else {
            if ((rc = pass_f32(ix, s, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, nf, k, kp, d_rowmask, ix->cur->w2_outi.p,
                               ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_flags.p, ix->cur->w2_flags.p + nf, d_status)))
                return rc;
            vdb::launch_scatter_results(ix->cur->w2_outi.p, ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_qidx.p, nf, (uint32_t)k,
                                        d_out_ids, d_out_dists, d_out_counts, s);
        }
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> f2(2 * (size_t)nf + 4);
        HIP_TRY(hipMemcpyAsync(f2.data(), ix->cur->w2_flags.p, 2 * (size_t)nf * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(f2.data() + 2 * (size_t)nf, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        status |= f2[2 * (size_t)nf];
        if (status & vdb::ST_ZERO_QUERY)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
        std::vector<uint32_t> todo2;
        for (uint32_t j = 0; j < nf; ++j) {
            bool cert = f2[j] != 0, ovf = f2[nf + j] != 0;
            if (ovf) ++ix->cur->stats[2];
            if (cert && !ovf && !force_exact) continue;
            todo2.push_back(todo[j]);
        }
        todo.swap(todo2);
    }
    // ---- exact fallback for the queries the MFMA tiers could not certify.  Up to 8 of them share one pass
    // over the rows; a row survives for a query only if its exact distance is <= the k-th exact distance
    // the re-rank already found (a valid upper bound), so each query is left with a handful of keys.
    uint32_t n_fallback = 0;
    n_fallback = (uint32_t)todo.size();
    if (!todo.empty()) {
        const uint32_t cap = 32768;
        if ((rc = ensure_ranks(ix))) return rc;
        if ((rc = ix->cur->w_exact.ensure(std::max<size_t>((size_t)8 * cap, n)))) return rc;
        if ((rc = ix->cur->w_exsel.ensure((size_t)8 * MAX_SELECT + 8))) return rc;
        uint32_t* d_cnt8 = ix->cur->w_cnt.p + 3 * SUPER;            // [8] survivors per query, [8..16) select counts
        std::vector<uint32_t> dense;                             // queries whose bounded pass overflowed
        for (size_t g0 = 0; g0 < todo.size(); g0 += 8) {
            const uint32_t nqf = (uint32_t)std::min<size_t>(8, todo.size() - g0);
            HIP_TRY(hipMemsetAsync(d_cnt8, 0, 16 * 4, s));
            vdb::ExactMultiParams ep{};
            ep.rows = ix->d_rows; ep.ld = ld; ep.dim = ix->dim; ep.n_rows = n; ep.qp = ix->cur->w_qp.p; ep.qnorm = ix->cur->w_qnorm.p;
            ep.nd = ix->d_nd; ep.rowmask = d_rowmask; ep.idrank = ix->ids_monotone ? nullptr : ix->d_idrank.p;
            ep.metric = ix->metric; ep.nqf = nqf;
            for (uint32_t j = 0; j < nqf; ++j) ep.qidx[j] = todo[g0 + j];
            ep.prev_dists = d_out_dists; ep.prev_counts = d_out_counts; ep.k = (uint32_t)k;
            ep.keys = ix->cur->w_exact.p; ep.cap = cap; ep.cnt = d_cnt8; ep.status = d_status;
            vdb::launch_exact_multi(ep, s);
            uint32_t h_cnt[8];
            HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt8, nqf * 4, hipMemcpyDeviceToHost, s));
            vdb::SelectParams sp{};
            sp.keys = ix->cur->w_exact.p; sp.stride = cap; sp.counts = d_cnt8; sp.n_fixed = 0; sp.cap = cap; sp.kk = (uint32_t)k;
            sp.out_keys = ix->cur->w_exsel.p; sp.out_stride = MAX_SELECT; sp.out_cnt = d_cnt8 + 8;
            vdb::launch_select(sp, nqf, s);
            vdb::EmitMultiParams em{};
            em.keys = ix->cur->w_exsel.p; em.key_stride = MAX_SELECT; em.cnt = d_cnt8 + 8;
            em.rank2row = ix->ids_monotone ? nullptr : ix->d_rank2row.p; em.row_ids = ix->d_row_ids;
            em.out_ids = d_out_ids; em.out_dists = d_out_dists; em.out_count = d_out_counts; em.k = (uint32_t)k; em.nqf = nqf;
            for (uint32_t j = 0; j < nqf; ++j) em.qidx[j] = todo[g0 + j];
            vdb::launch_emit_multi(em, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            for (uint32_t j = 0; j < nqf; ++j)
                if (h_cnt[j] > cap) dense.push_back(todo[g0 + j]);   // e.g. every row ties with the bound
        }
        for (uint32_t q : dense)
            if ((rc = exact_one(ix, s, q, k, d_rowmask, d_out_ids + (size_t)q * k, d_out_dists + (size_t)q * k,
                                d_out_counts + q)))
                return rc;
    }
    ix->cur->stats[12] = since();                       // whole call, ns
    ix->cur->stats[0] = nq32 - n_fallback;
    ix->cur->stats[1] = n_fallback;
    uint32_t st2 = status;
    if (n_fallback) {
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        st2 |= ix->cur->h_flags[0];
    }
    // status bits, the summary word, or a query norm beyond the kernels' no-NaN domain (word 2 is a running maximum: tame
    // values may stay, a wild one must not outlive its search)
    ix->cur->status_dirty = st2 != 0 || ix->cur->stats[6] != 0 || ix->cur->stats[2] != 0 || ix->cur->h_flags[2] > 0x53800000u;
    if (st2 & vdb::ST_NAN)
        return fail(VDB_ERR_NAN, "NaN distance (the reference panics here, flat_index.rs:62)");
    return VDB_OK;
}

void publish_stats(Index* ix) { memcpy(ix->stats, ix->cur->stats, sizeof(ix->stats)); }

// searches submitted and not yet waited for (vdb_flat_search_batch_device_submit): the row store must not change under them
bool in_flight(const Index* ix) { return ix->wsv && (ix->wsv[0].busy || ix->wsv[1].busy); }
int refuse_in_flight() { return fail(VDB_ERR_INVALID_ARGUMENT, "a submitted search is still in flight on this handle: wait for it first"); }

int search_device(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                  size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                  hipStream_t user_stream) {
    // a synchronous search takes a context no submitted search is using
    ix->cur = ix->wsv[0].busy ? &ix->wsv[1] : &ix->wsv[0];
    if (ix->cur->busy) return fail(VDB_ERR_INVALID_ARGUMENT, "two submitted searches are in flight on this handle: wait for one first");
    // (the other workspace may serve the alternating passes of a large batch: the handle mutex is held until part 2 is done,
    // so no submit can claim it meanwhile)
    int rc = search_part1(ix, d_q, nq, dim, k, d_idmask, mask_bits, d_out_ids, d_out_dists, d_out_counts, user_stream, !in_flight(ix));
    if (rc) { ix->cur->ctx.pending = false; publish_stats(ix); ix->cur = &ix->wsv[0]; return rc; }
    rc = search_part2(ix, nullptr);
    publish_stats(ix);
    ix->cur = &ix->wsv[0];
    return rc;
}

}  // namespace vdbi
