// kernels.h -- parameter blocks and launchers shared by the HIP translation units
// and the host-side index (vdb_flat.cpp).  gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vdb {

enum Metric : int { EUCLID = 0, COSINE = 1, DOT = 2 };

// Status bits written by kernels into the per-search status word.
enum : uint32_t {
    ST_NAN = 1u,          // a NaN distance was produced (reference: panic, flat_index.rs:62)
    ST_ZERO_QUERY = 2u,   // zero-norm query under Cosine (distance.rs:51-55)
};

constexpr uint64_t EMPTY_KEY = ~0ull;   // pool / candidate slot that holds no row
constexpr int KSTAGE = 32;              // K elements per LDS stage; device row stride is a multiple of it

// Order-preserving map f32 -> u32 (ascending unsigned == ascending float; -0 < +0).
__host__ __device__ inline uint32_t f32_to_ordered(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t b = __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; uint32_t b = c.u;
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float ordered_to_f32(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    union { float f; uint32_t u; } c; c.u = b; return c.f;
#endif
}
// Candidate key: ranking score in the high word, device row (or id rank) in the low word.
// A NaN score gets the smallest key so that it is always selected and reported.
__host__ __device__ inline uint64_t make_key(float score, uint32_t row) {
    uint32_t hi = (score != score) ? 0u : f32_to_ordered(score);
    return ((uint64_t)hi << 32) | row;
}

// Raw pool entry written by the fused kernels' epilogue: the score's f32 bits and the row (0xffffffff =
// ineligible row).  The select's gather turns it into an ordered candidate key (cheaper than doing the
// order-preserving transform and the NaN test inside the MFMA kernel, where every VALU instruction
// takes cycles from the f32 MFMA pipe).
__host__ __device__ inline uint64_t make_raw_key(float score, uint32_t row) {
#ifdef __HIP_DEVICE_COMPILE__
    return ((uint64_t)__float_as_uint(score) << 32) | row;
#else
    union { float f; uint32_t u; } c; c.f = score; return ((uint64_t)c.u << 32) | row;
#endif
}
__device__ inline uint64_t raw_to_key(uint64_t raw) {
    const uint32_t row = (uint32_t)raw;
    return row == 0xffffffffu ? EMPTY_KEY : make_key(__uint_as_float((uint32_t)(raw >> 32)), row);
}

// ---------------------------------------------------------------- row store statistics
struct RowStatsParams {
    const float* rows; uint32_t ld; uint32_t dim;
    uint32_t row_begin, row_end;
    int metric;
    float* nd;        // exact-order norm  sqrt(fold(x*x))      (vector.rs:35-37)
    float* alpha;     // score = fma(dot, alpha, beta)
    float* beta;
    uint32_t* nd2max_bits;   // atomicMax of the f32 bits of fold(x*x); [2] / [3]: max of the bf16 rounding error of a
                             // row, as f32 bits of |x - bf16(x)|^2 (slot 2) and of |x - bf16(x)|^2 / |x|^2 (slot 3);
                             // [4]: atomicMin of ~bits of the smallest positive finite margin (stored complemented so that 0 = none yet)
    // Screening-tier certificate, LOCAL form (Dot / Euclid; null under Cosine, whose relative row error is bounded by
    // 2^-9 whatever the data): margin[row] = max(m_e |x - bf16(x)| + m_n |x|, m_b |x|), rounded up, so that
    // g_q * margin[row] >= the whole row-dependent part of the score error for a query with g_q = |q| + kappa |q - bf16(q)|
    // (QueryPrepParams::kappa, m_b = B/kappa); beta_shrink: Euclid's beta = |x|^2 (1 - beta_shrink), rounded down.
    float* margin; float m_e, m_n, m_b, beta_shrink;
};
void launch_row_stats(const RowStatsParams& p, hipStream_t s);
// bf16 shadow of rows [row_begin, row_end): v_cvt_pk_bf16_f32 (RNE) of every element of the padded row -- the rounding the
// f32-row screening kernels apply to their fragments in registers
void launch_rows_to_bf16(const float* rows, uint16_t* rows16, uint32_t ld, uint32_t row_begin, uint32_t row_end, hipStream_t s);
// compact bf16 copy of the S = 2^shift sample rows of the screening tier: out[j] = bf16(rows[sample_row(j)]), j < S, the
// mapping of the sample kernels (position (j & 255) * (S >> 8) + (j >> 8), row = (position * n_rows) >> shift)
void launch_sample_to_bf16(const float* rows, uint32_t ld, uint32_t n_rows, uint32_t n_sample, uint32_t shift, uint16_t* out, hipStream_t s);

// count live rows whose norm is exactly zero (Cosine: distance.rs:51-55)
void launch_count_zero_live(const float* nd, const uint32_t* livemask, uint32_t n_rows,
                            uint32_t* out_count, hipStream_t s);

// rowmask[w] = live[w] & (id mask bits gathered through row_ids)
void launch_build_rowmask(const uint64_t* row_ids, const uint32_t* livemask, const uint64_t* idmask,
                          uint64_t mask_bits, uint32_t n_rows, uint32_t* out_mask, hipStream_t s);

// ---------------------------------------------------------------- query preparation
struct QueryPrepParams {
    const float* q_in; uint32_t dim; uint32_t nq;      // [nq][dim] as handed over
    float* qp; uint32_t ld; uint32_t nq_pad;           // [nq_pad][ld], zero padded
    float* qnorm;                                      // exact-order norm per query [nq_pad]
    float* thr;                                        // padding queries get -inf here (nothing passes)
    int metric;
    uint32_t* status;
    uint16_t* qb;                                      // may be null: [nq_pad][ld] bf16 (RNE) copy for the screening tier
    float* qerr;                                       // with qb: |q - bf16(q)| per query (upper bound)
    float* qg; float kappa;                            // with qb (may be null): g_q = |q| + kappa |q - bf16(q)|, rounded up (RowStatsParams::margin)
    uint32_t* clear_a; uint32_t* clear_b;              // may be null: per-query flag words this kernel zeroes (cert, overflow)
};
void launch_query_prep(const QueryPrepParams& p, hipStream_t s);

// ---------------------------------------------------------------- dense scores (sample / small N)
struct DenseParams {
    const float* rows; uint32_t ld; uint32_t n_rows;
    const float* qp; uint32_t nq_pad;                  // multiple of 32
    const float* alpha; const float* beta;
    const uint32_t* rowmask;                           // may be null (all rows eligible)
    uint32_t n_sample;                                 // sample j -> row j*n_rows/n_sample
    uint64_t* keys; uint32_t key_stride;               // keys[q*key_stride + j]
};
void launch_dense_scores(const DenseParams& p, hipStream_t s);

// ---------------------------------------------------------------- per-query radix select
struct SelectParams {
    const uint64_t* keys; size_t stride;               // per-query key block
    const uint32_t* counts; uint32_t n_fixed;          // counts != null: n = min(counts[q], cap)
    uint32_t cap;
    // gather mode (n_sub > 0): the query's keys live in n_sub sub-pools of capacity capl,
    // keys[(q*n_sub + s)*capl + j], j < sub_counts[q*n_sub + s]
    const uint32_t* sub_counts; uint32_t n_sub; uint32_t capl;
    uint32_t wg_major;                                 // 1: sub-pool i = wg*4 + r of query q has its count at (wg*256 + q)*4 + r and its keys at that * capl (bf16 tier)
    size_t blk_keys, blk_cnts;                         // wg_major, more than 256 queries in one launch: query q lives in block q >> 8, whose pools start blk_keys keys / blk_cnts counts after the previous block's
    uint32_t kk;                                       // how many smallest keys to keep (<= 2048)
    uint64_t* out_keys; uint32_t out_stride;           // sorted ascending, padded with EMPTY_KEY
    uint32_t* out_cnt;
    float* out_thr;                                    // may be null: score of the kk-th key, +inf if fewer
    const float* shift_g; const uint32_t* shift_m_bits; // may be null: out_thr[q] -= shift_g[q] * f32(*shift_m_bits)  (plain-score
                                                       // sample threshold -> lower-bound units, see launch_sample_bf16)
    uint32_t* ovf;                                     // may be null: set when counts[q] > cap
    uint32_t* summary;                                 // may be null: OR-ed with 2 whenever an overflow flag is set
    uint32_t flag_truncation;                          // 1: more valid keys than kk also sets ovf[q] (the caller needs ALL of them)
    const uint64_t* lo_excl;                           // may be null: only keys > lo_excl[q] take part (chunked large k)
    uint64_t* out_last;                                // may be null: largest selected key per query (unchanged if none)
    // EMIT (may be null; keys = ordered(exact distance) << 32 | id rank, as the exact scans write them): the selected keys are
    // also written as final results -- emit_ids / emit_dists [q * emit_stride + i], emit_counts[q] -- and the search's status
    // word is copied to emit_status[q].  The outputs may live in mapped host memory (the small-index direct path).
    uint64_t* emit_ids; float* emit_dists; uint32_t* emit_counts; uint32_t emit_stride;
    const uint32_t* emit_rank2row; const uint64_t* emit_row_ids;
    const uint32_t* emit_status_in; uint32_t* emit_status;
};
void launch_select(const SelectParams& p, uint32_t nq, hipStream_t s);
// The screening tier's THRESHOLD select: only the score of the kk-th smallest of n_fixed <= 4096 keys per query is wanted
// (out_thr, with the shift_g / shift_m_bits adjustment) -- no sorted list.  A 256-thread workgroup keeps the keys' score
// words in registers and finds the value with four 8-bit radix passes: ~5 us against the general kernel's ~15.
void launch_thr_select(const SelectParams& p, uint32_t nq, hipStream_t s);

// ---------------------------------------------------------------- fused MFMA score + threshold filter
struct FusedParams {
    const float* rows; uint32_t ld; uint32_t n_rows;
    const float* qp;                                   // padded queries, row q_base is the first of this launch
    uint32_t q_base;
    const float* alpha; const float* beta;
    const uint32_t* rowmask;                           // NEVER null here: the live mask when there is no filter
    const float* thr;                                  // [nq_pad] inclusive threshold per query
    // candidate pools: one private sub-pool per (query, row range, row part, lane half):
    //   sub = ((q*n_wg + range)*RP + part)*2 + half ; keys at pool[sub*capl ..], count at pool_cnt[sub]
    uint64_t* pool; uint32_t* pool_cnt; uint32_t capl;
    uint32_t n_wg;                                     // row ranges = grid.x
    uint32_t ablate;                                   // diagnostics only (VDB_FUSED_ABLATE): 1 skip MFMAs, 2 skip staging, 4 skip barriers, 8 skip epilogue
};
// nqt = number of 32-query tiles handled per workgroup (1, 2, 4 or 8); grid.y super-tiles of 32*nqt queries
void launch_fused(const FusedParams& p, int nqt, uint32_t n_super, hipStream_t s);
size_t fused_lds_bytes(int nqt);
uint32_t fused_tile_rows(int nqt);
uint32_t fused_subpools_per_query(int nqt, uint32_t n_wg);
#ifdef VDB_DIAG
// 2-image LDS-DMA variant of the headline shape (nqt = 8; diagnostics build: A/B against dma3)
void launch_fused_dma(const FusedParams& p, uint32_t n_super, hipStream_t s);
#endif
// three-image ring, barrier in the middle of a stage (kernels_fused_dma3.hip)
void launch_fused_dma3(const FusedParams& p, uint32_t n_super, hipStream_t s);

// ---------------------------------------------------------------- bf16 screening tier (kernels_fused_bf16.hip)
struct FusedBf16Params {
    const float* rows; uint32_t ld; uint32_t n_rows;
    const uint16_t* rows16;                            // bf16 shadow of `rows`, same pitch (kernels_fused_s16.hip; null unless vdb_flat_set_shadow and ld % 64 == 0)
    const uint16_t* qb;                                // [256][ld] bf16 queries of this pass (zero padded)
    const float* alpha; const float* beta;
    // non-null (Dot / Euclid): the kernels rank by the LOWER-BOUND score  fma(-qg[q], margin[row], fma(dot, alpha, beta)),
    // i.e. the score minus everything the bf16 rounding, the MFMA accumulation and the oracle's own f32 fold can
    // contribute for THIS row and query -- the certificate then needs no per-index maxima (kernels_aux.hip cert_test)
    const float* margin; const float* qg;
    const uint32_t* rowmask;                           // NEVER null: the live mask when there is no filter
    // "No score of this launch can be NaN" (fused_no_nan below): the index scalars (max |d|^2, smallest positive |d|^2 under
    // Cosine) and the largest query norm of the search as f32 bits (query_prep: status block word 2).  May be null (= unknown).
    const uint32_t* scalars; const uint32_t* qmax_bits;
    // filter mode: keys with score <= thr[q] go to the private sub-pool
    //   sub = ((wg*256 + q)*2 + row half)*2 + lane half ; keys at pool[sub*capl ..], count at pool_cnt[sub]  (workgroup-major)
    const float* thr;                                  // [256]
    uint64_t* pool; uint32_t* pool_cnt; uint32_t capl;
    uint32_t n_wg;                                     // row ranges = grid.x
    // the WIDE filter kernel (kernels_fused_bf16w.hip) serves two consecutive 256-query blocks per launch: qb / thr / qg cover
    // 512 queries, block b's pools start at pool + b * pool_block_stride (keys) and pool_cnt + b * cnt_block_stride
    size_t pool_block_stride, cnt_block_stride;
    // sample mode: n_sample = 2^sample_shift <= n_rows, sample j -> row (j*n_rows) >> sample_shift; per query and
    // group of 64 sample rows the smallest key
    uint32_t n_sample, sample_shift, sample_block; uint64_t* minkeys; uint32_t minkey_stride;   // minkeys[q*minkey_stride + group]
    uint32_t ablate;                                   // diagnostics only (VDB_BF16_ABLATE): 1 skip LDS reads + MFMAs (unpipelined kernel only), 2 skip the row DMA, 4 skip the query DMA, 8 skip the epilogue, 16 (pipelined kernel) thresholds = -inf: nothing passes the filter
};
#ifdef VDB_DIAG
void launch_fused_bf16(const FusedBf16Params& p, hipStream_t s);      // unpipelined filter pass (diagnostics build: A/B against bf16p)
#endif
void launch_sample_bf16(const FusedBf16Params& p, uint32_t n_cu, hipStream_t s);
void launch_fused_bf16p(const FusedBf16Params& p, hipStream_t s);     // kernels_fused_bf16p.hip: the filter pass, software-pipelined (default)
void launch_fused_bf16w(const FusedBf16Params& p, hipStream_t s);     // kernels_fused_bf16w.hip: the WIDE filter pass, 128 rows x 512 queries per workgroup (batches above 256 queries)
uint32_t fused_bf16w_tile_rows();
// With every row norm and query norm in [2^-40, 2^40] (zero allowed) no accumulator can overflow and no alpha / beta is
// non-finite, so fma(acc, alpha, beta) is never NaN -- the filter epilogues may then test the MINIMUM of four scores against the
// threshold (v_min_f32 drops a NaN operand; a NaN score must pass the filter, flat_index.rs:62).  Otherwise they add a NaN
// test per group.  Wave-uniform, evaluated once per launch.
__device__ inline bool fused_no_nan(const uint32_t* scalars, const uint32_t* qmax_bits, bool cosine) {
    if (!scalars || !qmax_bits) return false;
    const uint32_t n2max = scalars[0], qmax = *qmax_bits;
    bool ok = n2max <= 0x67800000u /* 2^80 */ && qmax <= 0x53800000u /* 2^40 */;
    if (cosine) { const uint32_t mb = scalars[5]; ok = ok && mb != 0u && (~mb) >= 0x17800000u /* 2^-80 */; }
    return ok;
}
void launch_fused_s16(const FusedBf16Params& p, hipStream_t s);       // kernels_fused_s16.hip: the filter pass over p.rows16 (ld % 64 == 0)
void launch_sample_s16(const FusedBf16Params& p, hipStream_t s);      // the sample pass over a COMPACT bf16 copy of the sample rows (p.rows16 = the copy)
uint32_t fused_bf16_tile_rows();
uint32_t fused_bf16_subpools_per_query(uint32_t n_wg);
uint32_t fused_bf16_sample_groups(uint32_t n_sample);

// compact re-run of uncertified queries: gather padded query rows / norms, scatter results back
void launch_gather_queries(const float* qp, const float* qnorm, uint32_t ld, const uint32_t* qidx, uint32_t n,
                           uint32_t n_pad, float* qp_out, float* qnorm_out, float* thr_out, hipStream_t s);
void launch_scatter_results(const uint64_t* ids, const float* dists, const uint32_t* counts, const uint32_t* qidx,
                            uint32_t n, uint32_t k, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                            hipStream_t s);

// the same for an explicit (source, destination) list
void launch_scatter_results_list(const uint64_t* ids, const float* dists, const uint32_t* counts, const uint32_t* src,
                                 const uint32_t* dst, uint32_t n, uint32_t k, uint64_t* out_ids, float* out_dists,
                                 uint32_t* out_counts, hipStream_t s);

// ---------------------------------------------------------------- exact re-rank + certification
struct RerankParams {
    const float* rows; uint32_t ld; uint32_t dim; uint32_t n_rows;
    const float* qp; const float* qnorm;
    const float* nd;
    const uint64_t* row_ids;
    const uint32_t* rowmask;                           // may be null
    const uint64_t* cand; uint32_t cand_stride; const uint32_t* cand_cnt; uint32_t kp;   // kp <= 256 candidates, sorted by score
    uint32_t kp_first, kp_step;                        // adaptive depth: re-rank kp_first, then kp_step more per round (0: kp at once)
    uint32_t* depth;                                   // may be null: candidates re-ranked per query (diagnostics)
    float* thr_next;                                   // may be null: for an UNCERTIFIED query the score cut above which no row can
                                                       // enter the top k (from the k-th exact distance found so far); NaN: no cut known
    int metric;
    uint32_t k;                                        // results wanted per query
    float eps_coef; const uint32_t* nd2max_bits;       // certification bound inputs (max row norm^2, f32 bits)
    uint64_t* out_ids; float* out_dists; uint32_t* out_counts; uint32_t out_stride;
    uint32_t* cert;                                    // [nq]: 1 = certified exact
    uint32_t* status;
    const float* thr;                                  // may be null: the filter threshold the candidates passed; a short
                                                       // candidate list under a FINITE threshold is never certified
    const float* qerr; float c_acc;                    // bf16 screening tier (qerr != null): |q - bf16(q)| per query and the
                                                       // MFMA accumulation coefficient; the row-side error maxima are
                                                       // nd2max_bits[2] and [3]
    uint32_t lb_scores;                                // 1 (with qerr; Dot / Euclid): candidate scores and thresholds are the
                                                       // LOWER-BOUND scores of FusedBf16Params::margin -- every row-dependent
                                                       // error term is already inside them
    uint32_t lds_row_stride, lds_chunk;                // filled by launch_rerank
};
void launch_rerank(const RerankParams& p, uint32_t nq, hipStream_t s);
// exhaustive variant: EVERY candidate of the list (up to cand_stride, any order) is re-ranked, the best k are kept;
// cert[q] = 1 unless the list was truncated upstream (the caller's overflow flag) or a NaN score was seen
void launch_rerank_all(const RerankParams& p, uint32_t nq, hipStream_t s);

// ---------------------------------------------------------------- direct exact scan of a SMALL index for a few queries
// Index::search as it stands (flat_index.rs:52-65) for indexes of at most 16384 rows and batches of at most 8 queries -- the shape
// of BASELINE configs[0] (10k x 128, ONE query): every row's exact distance in the reference's operation order, straight from
// the raw queries (which may live in mapped host memory): no query preparation pass, no MFMA scores, no certificate.
struct SmallScanParams {
    const float* rows; uint32_t ld; uint32_t dim; uint32_t n_rows;
    const float* q_in; uint32_t nq;                    // [nq][dim] as handed over
    const float* nd; const uint32_t* rowmask; const uint32_t* idrank; int metric;
    uint64_t* keys; uint32_t key_stride;               // key = ordered(dist) << 32 | id rank, EMPTY if ineligible
    // keep == 0: keys[q * key_stride + row], every row's key.  keep = k' < 256: every workgroup of 256 rows keeps only its k'
    // smallest keys (any global top-k with k <= k' is among them): keys[q * key_stride + workgroup * k' + i], EMPTY padded --
    // the select then sees n/256 * k' keys instead of n (10k rows, k = 10: 400 keys, ranked by counting in ~3 us instead of
    // seven radix passes over 10k keys in 57 us)
    uint32_t keep;
    uint32_t* status;                                  // ST_NAN / ST_ZERO_QUERY
};
void launch_small_scan(const SmallScanParams& p, hipStream_t s);
uint32_t small_scan_groups(uint32_t n_rows);           // workgroups along the rows

// ---------------------------------------------------------------- exact scan (fallback, any k)
struct ExactScanParams {
    const float* rows; uint32_t ld; uint32_t dim; uint32_t n_rows;
    const float* q; const float* qnorm;                // one padded query row and its exact-order norm
    const float* nd;
    const uint32_t* rowmask;
    const uint32_t* idrank;                            // may be null: rank == row
    int metric;
    uint64_t* keys;                                    // [n_rows]: ordered(exact dist)<<32 | idrank, EMPTY if ineligible
    uint32_t* status;
};
void launch_exact_scan(const ExactScanParams& p, hipStream_t s);

// Bounded exact scan for up to 8 uncertified queries in ONE pass over the rows: a row is kept for query j
// only if its exact distance is <= bound_j, the k-th exact distance the re-rank already produced (an upper
// bound of the true k-th distance), so the survivors are a handful of keys per query.
struct ExactMultiParams {
    const float* rows; uint32_t ld; uint32_t dim; uint32_t n_rows;
    const float* qp; const float* qnorm;               // padded query block and norms of the whole batch
    const float* nd; const uint32_t* rowmask; const uint32_t* idrank;
    int metric;
    uint32_t nqf; uint32_t qidx[8];                    // batch indices of the queries of this pass
    const float* prev_dists; const uint32_t* prev_counts; uint32_t k;   // re-rank outputs: bound = dists[q*k + k-1] if counts[q] == k
    uint64_t* keys; uint32_t cap; uint32_t* cnt;       // keys[j*cap + slot], cnt[j] (may exceed cap: overflow)
    uint32_t* status;
};
void launch_exact_multi(const ExactMultiParams& p, hipStream_t s);

struct EmitParams {                                    // sorted exact keys -> (id, dist) outputs
    const uint64_t* keys; uint32_t cnt_max; const uint32_t* cnt;
    const uint32_t* rank2row; const uint64_t* row_ids;
    uint64_t* out_ids; float* out_dists; uint32_t* out_count; uint32_t k;
    uint32_t accumulate;                               // 1: *out_count += n (chunked large k), 0: *out_count = n
};
// emit for several queries at once: query j (= blockIdx.y) reads keys + j*key_stride, cnt[j] and writes
// to out_*[qidx[j]*k ..] / out_count[qidx[j]]
struct EmitMultiParams {
    const uint64_t* keys; uint32_t key_stride; const uint32_t* cnt;
    const uint32_t* rank2row; const uint64_t* row_ids;
    uint64_t* out_ids; float* out_dists; uint32_t* out_count; uint32_t k;
    uint32_t nqf; uint32_t qidx[8];
};
void launch_emit_multi(const EmitMultiParams& p, hipStream_t s);
void launch_emit(const EmitParams& p, hipStream_t s);

// ---------------------------------------------------------------- multi-GPU partial merge
void launch_merge_parts(const uint64_t* ids, const float* dists, const uint32_t* counts, uint32_t nparts,
                        uint32_t nq, uint32_t k, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                        hipStream_t s);

// ---------------------------------------------------------------- candidate-list distances (HNSW offload hook)
struct PairDistParams {
    const float* rows; uint32_t ld; uint32_t dim;
    const float* qp; const float* qnorm; const float* nd;
    const uint32_t* pair_query; const uint32_t* pair_row; uint32_t n_pairs;   // pair_row 0xffffffff = unknown id
    int metric;
    float* out; uint32_t* status;                                            // ST_NAN / ST_ZERO_QUERY (zero norm on either side)
};
void launch_pair_distances(const PairDistParams& p, hipStream_t s);
// HNSW hooks: zero-norm Cosine pairs are written as the NaN bit pattern `mark` (status is not touched)
struct PairEvalParams {
    const float* rows; uint32_t ld; uint32_t dim; const float* nd;
    const float* qp; const float* qnorm;                 // prepared queries (query side of mode 0 / 2)
    const uint32_t* a; const uint32_t* b; uint32_t n;    // mode 0: (query a[i], row b[i]); mode 1: (row a[i], row b[i]); mode 2: (query q0, row i)
    uint32_t q0; int mode; int metric; uint32_t mark;
    float* out;
};
void launch_pair_eval(const PairEvalParams& p, hipStream_t s);
// HNSW build: exact reference distances of up to 16 STORED rows (the vectors being inserted) against device rows
// [0, n_scan) in ONE pass over the rows -- out[j * ldm + r] = distance(row qrow[j], row r); zero-norm Cosine pairs are
// written as `mark`.  `out` may live in mapped host memory.
struct ScanRowsParams {
    const float* rows; uint32_t ld; uint32_t dim; const float* nd; int metric; uint32_t mark;
    uint32_t nq; uint32_t qrow[16];
    uint32_t n_scan; float* out; size_t ldm;
};
void launch_scan_rows(const ScanRowsParams& p, hipStream_t s);

// ---------------------------------------------------------------- certificate diagnostics (vdb_flat_debug_*)
// dense[q*n_rows + row] = score bits of every key the filter pass wrote for query q (wg-major pools of the bf16 tier)
void launch_pool_to_dense(const uint64_t* pool, const uint32_t* pool_cnt, uint32_t n_sub, uint32_t capl, uint32_t nq,
                          uint32_t n_rows, float* dense, hipStream_t s);
// out[i] = cert_test(p, qi[i], T[i], ek[i]) -- the PRODUCTION certification function of rerank_kernel, probed directly
void launch_cert_probe(const RerankParams& p, const uint32_t* qi, const float* T, const float* ek, uint32_t n, uint32_t* out,
                       hipStream_t s);

// *code = VDB_PENDING_HOST (100) when the status block of a search (flags[0] status bits, flags[1] summary of
// uncertified / overflowed queries) is non-zero, else 0
void launch_write_code(const uint32_t* flags, int32_t* code, hipStream_t s);

// ---------------------------------------------------------------- device-resident HNSW search (kernels_hnsw.hip)
struct HnswSearchParams {
    const float* rows; uint32_t ld; uint32_t dim; const float* nd; int metric;
    const float* qp; const float* qnorm;                 // prepared queries
    // the graph of vdb_hnsw.cpp mirrored in HBM, indexed by node id
    const uint32_t* row_of; uint32_t n_ids;                              // row_of = 0xffffffff: absent / deleted
    // lists are padded with 0xffffffff; *_row holds the device row of each listed neighbour (0xffffffff = deleted)
    const uint32_t* nbr0; const uint32_t* nbr0_row; uint32_t stride0;    // layer 0 lists
    const uint32_t* up_off; const uint32_t* nbrU; const uint32_t* nbrU_row; uint32_t strideU;   // list (id, l >= 1) = up_off[id] + l - 1
    const uint32_t* level; const uint32_t* cnt0; const uint32_t* cntU;   // (unused by the kernel; kept for inspection)
    uint32_t entry_point, max_level, ef, k;
    uint64_t* out_ids; float* out_dists; uint32_t* out_counts;           // [nq][k]
    uint32_t* fail;                                                      // [nq]: 1 = structures overflowed, redo on the host
    uint32_t* status;
    // INSERT WALKS (vdb_hnsw.cpp build): the "query" is a STORED row (qrow[q], its norm nd[qrow[q]]) that the graph does not hold
    // yet, the walk is insert()'s (graph.rs:262-297): ef = 1 above the node's level qlevel[q], ef at and below it, and every
    // distance it evaluates is RECORDED -- rec_row (the evaluated node's ID) / rec_d [q * rec_cap + i], rec_cnt[q] entries (may exceed rec_cap: the rest
    // was dropped) -- for the host's authoritative replay.  All null in a search.  A zero-norm Cosine pair is recorded as
    // rec_zero_mark and ends the walk.
    const uint32_t* qrow; const uint32_t* qlevel;
    uint32_t* rec_row; float* rec_d; uint32_t* rec_cnt; uint32_t rec_cap; uint32_t rec_zero_mark;
    uint32_t chunk, stage_rows;                                          // set by launch_hnsw_search: staging chunk (elements), rows per staging buffer
};
void launch_hnsw_search(const HnswSearchParams& p, uint32_t nq, hipStream_t s);
// incremental update of the graph mirror: n0 layer-0 records [id, row, level, up_off, ids[stride0], rows[stride0]] and nU
// upper-list records [list index, ids[strideU], rows[strideU]] (uint32 words), scattered into the mirror arrays
struct HnswScatterParams {
    const uint32_t* rec0; uint32_t n0; const uint32_t* recU; uint32_t nU;
    uint32_t* row_of; uint32_t* level; uint32_t* up_off; uint32_t* nbr0; uint32_t* nbr0_row; uint32_t stride0;
    uint32_t* nbrU; uint32_t* nbrU_row; uint32_t strideU;
};
void launch_hnsw_scatter(const HnswScatterParams& p, hipStream_t s);
bool hnsw_search_supported(uint32_t dim, uint32_t ef, uint32_t k, uint32_t max_list);

void launch_merge_packed(const int32_t* packed, size_t words_per_part, uint32_t nparts, uint32_t nq, uint32_t k,
                         uint64_t* out_ids, float* out_dists, uint32_t* out_counts, uint32_t* out_status, hipStream_t s);

}  // namespace vdb
