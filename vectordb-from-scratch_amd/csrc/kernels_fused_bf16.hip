// kernels_fused_bf16.hip -- the SCREENING tier of the search: ranking scores of every row against up to 256
// queries on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32-input MFMA rate), so that the pass
// over the f32 rows is bound by HBM, not by arithmetic.  The scores only RANK rows; results stay the
// reference's exact f32 distances (distance.rs:37-73) because
//   * the candidates this kernel keeps are re-ranked with the reference's own f32 operation order, and
//   * the re-rank certifies, from a rigorous bound on the bf16 rounding error (|dot_bf16 - dot| <=
//     2^-8 |q||d| (1 + small), DESIGN.md "screening tier"), that no excluded row can enter the top k;
//     a query that cannot be certified is re-done by the f32 MFMA tier and, failing that, the exact scan.
//
// Rows stay f32 in HBM (the byte layout of persistence/mmap.rs:77-84).  A stage is 32 K-elements of
// 256 rows (f32, 32 KB) and of 256 queries (bf16, prepared once per batch by query_prep, 16 KB), brought
// into a 3-image LDS ring by LDS-DMA (global_load_lds_dwordx4).  Each of the 8 waves owns a 128-row x
// 64-query block of the 256 x 256 tile (4 x 2 MFMA tiles, 128 accumulator VGPRs): it reads its f32 row
// fragments from LDS, rounds them to bf16 in registers (v_cvt_pk_bf16_f32, RNE) and issues 16 MFMAs per
// stage.  Per stage and CU: 32 KB from HBM, 1024 MFMA cycles per SIMD, 160 KB of LDS reads -- all below the
// ~2500 cycles the HBM share of one CU needs for 32 KB, so the kernel is HBM-bound by construction.
//
// Two DMA stages are kept in flight: stage s+2 is issued at the top of stage s, and the wave waits with a
// COUNTED s_waitcnt (vmcnt <= the 6 pieces of stage s+1) before the barrier that publishes stage s.  Loads
// complete in order, so "at most 6 outstanding" implies every piece of stage s has landed whatever other
// loads or stores (pool appends) the wave issued since.
//
// The same kernel body runs in SAMPLE mode over S sample rows: instead of filtering by a threshold, every
// lane keeps the smallest (score,row) key of the 64 rows it owns per query ("group minimum").  The kp-th
// smallest of a query's group minima is an inclusive threshold that at least kp rows meet, and the
// instruction sequence per (row, query) is identical in both modes, so the scores agree bit for bit.
// NOTE: the FILTER pass runs by default in its software-pipelined form, kernels_fused_bf16p.hip (same tile, same DMA
// plan, same epilogue; one mid-stage barrier per stage); this file keeps the sample pass and the unpipelined filter
// pass (VDB_FUSED_PIPE=0).
#include "kernels.h"

#include <type_traits>

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NW = 8, NT = NW * 64;
constexpr int TR = 256;                          // rows per tile
constexpr int TQ = 256;                          // queries per tile
constexpr int A_ROWB = 128;                      // 32 f32 per row and stage
constexpr int B_ROWB = 64;                       // 32 bf16 per query and stage
constexpr int A_BYTES = TR * A_ROWB;             // 32 KB
constexpr int B_BYTES = TQ * B_ROWB;             // 16 KB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // 48 KB
constexpr int MT = 4, QT = 2;                    // MFMA tiles per wave: 4 x 32 rows, 2 x 32 queries

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ bf16x8 cvt8(const float4& lo, const float4& hi) {
    u32x4 r = {pk_bf16(lo.x, lo.y), pk_bf16(lo.z, lo.w), pk_bf16(hi.x, hi.y), pk_bf16(hi.z, hi.w)};
    return __builtin_bit_cast(bf16x8, r);
}
}  // namespace

// MARGIN: lower-bound scores for Dot / Euclid, exactly as in kernels_fused_bf16p.hip (same two fmas per element, so the
// sample's group minima and the filter pass agree bit for bit).
template <bool SAMPLE, bool MARGIN>
__global__ __launch_bounds__(NT, 2) void fused_bf16_kernel(FusedBf16Params p) {
    // three DISTINCT LDS objects, each access names its image at compile time (see kernels_fused_dma3.hip)
    __shared__ __attribute__((aligned(16))) char sImg0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg1[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg2[STAGE_BYTES];
    // per-row constants of a tile (alpha, beta, the row's eligibility-mask word), double buffered by tile parity;
    // filled by LDS-DMA as well, so that no wave ever holds a pending ordinary load inside the stage loop
    __shared__ __attribute__((aligned(16))) float sAlpha[SAMPLE ? TR : 2 * TR];            // (sample mode: one tile per workgroup, parity 0 only)
    __shared__ __attribute__((aligned(16))) float sBeta[SAMPLE ? TR : 2 * TR];
    __shared__ __attribute__((aligned(16))) uint32_t sMaskW[SAMPLE ? TR : 2 * TR];
    // sample mode: the device rows of the (single) tile's scattered sample rows live here instead of in four 64-bit address
    // registers per lane (with those the sample instance spilled in its last stages, and every scratch reload is a
    // vmcnt(0) -- a drain of the DMA pipeline); each lane reads back only what it wrote
    __shared__ uint32_t sRow[SAMPLE ? 32 * 64 : 1];
    __shared__ __attribute__((aligned(16))) float sMarg[MARGIN ? (SAMPLE ? TR : 2 * TR) : 4];
    __shared__ float sG[MARGIN ? TQ : 1];

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wr = w >> 2, wq = w & 3;                             // row half, query quarter of this wave
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t ld = p.ld;
    const uint32_t KS = ld / KSTAGE;

    // ---- the rows this workgroup covers
    // (sample mode: exactly ONE tile per workgroup, grid = number of sample tiles.  A compile-time tile count lets
    // the compiler drop the next-tile address state; with it the sample instance spilled registers to scratch, and
    // every scratch reload put a vmcnt(0) -- a full drain of the DMA pipeline -- into the stage loop)
    uint32_t r0 = 0, r1 = 0, ntiles_rt = 0;
    const uint32_t tile_first = SAMPLE ? blockIdx.x : 0u;
    constexpr uint32_t tile_step = 0;
    if (SAMPLE) {
        ntiles_rt = 1;
    } else {
        // row ranges in WHOLE tiles: a range of 15.26 tiles costs 16 tile iterations whatever its last tile holds, so
        // the tiles are dealt out whole -- some workgroups run one tile fewer, and the last tiles of the others meet
        // an HBM that is no longer contended
        const uint32_t nblk = (p.n_rows + TR - 1) / TR;
        const uint32_t b0 = (uint32_t)(((uint64_t)blockIdx.x * nblk) / p.n_wg);
        const uint32_t b1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * nblk) / p.n_wg);
        r0 = b0 * TR;
        r1 = (b1 * TR < p.n_rows) ? b1 * TR : p.n_rows;
        ntiles_rt = r0 < r1 ? (r1 - r0 + TR - 1) / TR : 0;
    }
    const uint32_t ntiles = SAMPLE ? 1u : ntiles_rt;
    // queries of this lane: one column in each of the wave's two 32-query MFMA tiles
    const uint32_t q_a = wq * 64 + c, q_b = q_a + 32;
    uint64_t* pool_a = nullptr; uint64_t* pool_b = nullptr;
    size_t sub_a = 0, sub_b = 0;
    float thr_a = 0.f, thr_b = 0.f;
    if (!SAMPLE) {
        sub_a = (((size_t)blockIdx.x * TQ + q_a) * 2 + wr) * 2 + h;          // counts workgroup-major too (kernels_fused_bf16p.hip)
        sub_b = (((size_t)blockIdx.x * TQ + q_b) * 2 + wr) * 2 + h;
        // The pool KEYS are laid out workgroup-major -- slot ((wg*256 + q)*4 + row half*2 + lane half)*capl -- so that the
        // few scattered appends of one workgroup fall into ONE 2 MB region instead of one region per query (256 regions
        // 2 MB apart: every append then missed the CU's address-translation cache in front of the row stream); the
        // select's gather knows both layouts (SelectParams::wg_major).
        pool_a = p.pool + ((((size_t)blockIdx.x * TQ + q_a) * 2 + wr) * 2 + h) * p.capl;
        pool_b = p.pool + ((((size_t)blockIdx.x * TQ + q_b) * 2 + wr) * 2 + h) * p.capl;
        thr_a = p.thr[q_a];
        thr_b = p.thr[q_b];
        // consume the two loads here: a first use inside the stage loop would get a compiler-inserted vmcnt(0)
        // there, i.e. a wait for every DMA in flight, once per tile
        asm volatile("" : "+v"(thr_a), "+v"(thr_b));
    }
    uint32_t pcnt_a = 0, pcnt_b = 0;
    if (ntiles == 0) {
        if (!SAMPLE) { p.pool_cnt[sub_a] = 0; p.pool_cnt[sub_b] = 0; }
        return;
    }
    const uint32_t total = ntiles * KS;
    const uint32_t last_row = p.n_rows - 1;
    // sample index -> device row.  The S sample positions are spread evenly over the rows ((pos * n) >> shift), and
    // CONSECUTIVE positions go to DIFFERENT tiles (index j = tile*256 + tile-row sits at position tile-row*tiles + tile):
    // when near neighbours are stored next to each other (data ordered by cluster) their sample rows then land in
    // different groups, each contributes its own group minimum, and the threshold stays as tight as on shuffled data
    // (with consecutive positions in one tile a 500-row cluster was represented by 4 minima, the threshold came from far
    // rows and thousands of keys overflowed the pools).  Block mode (sample_block != 0, diagnostics): tiles of
    // contiguous rows.
    auto sample_row_of = [&](uint32_t j) -> uint32_t {
        if (p.sample_block) return (j >> 8) * p.sample_block + (j & 255u);
        const uint32_t pos = (j & 255u) * (p.n_sample >> 8) + (j >> 8);
        return (uint32_t)(((uint64_t)pos * p.n_rows) >> p.sample_shift);
    };
    const char* __restrict__ rows_b = reinterpret_cast<const char*>(p.rows);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qb);

    // ---- DMA plan.  A stage image = 32 row pieces + 16 query pieces of 1 KB.  Wave w fills row pieces
    // 4w..4w+3 (8 rows x 128 B each: lane L -> row L>>3, 16-byte position L&7) and query pieces 2w, 2w+1
    // (16 queries x 64 B each: lane L -> query L>>2, position L&3).  Both images are XOR-swizzled so that the
    // fragment reads below are bank-conflict free: data chunk x of row r sits at position x ^ ((r>>1)&7),
    // data chunk x of query r at position x ^ ((r>>2)&3); the filling lane fetches the permuted source chunk.
    const uint32_t a_pr = lane >> 3, a_pp = lane & 7;
    const uint32_t b_pr = lane >> 2, b_pp = lane & 3;
    uint32_t a_chunk[4];                                                // source byte offset inside the 128-B stage
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t rt = 32 * w + 8 * j + a_pr;                      // row inside the tile
        a_chunk[j] = (a_pp ^ ((rt >> 1) & 7)) * 16;
    }
    // (the queries are stored by query_prep in exactly this image order, one 16 KB image per K stage: a wave's
    // query piece is 1 KB of CONTIGUOUS global memory -- 8 full 128-byte requests instead of 16 scattered 64-byte ones)
    const uint32_t ob[2] = {(2 * w) * 1024 + lane * 16, (2 * w + 1) * 1024 + lane * 16};
    (void)b_pr; (void)b_pp;
    const char* aptr[4];                                                // row pieces of the tile being fetched
    auto tile_rows_of = [&](uint32_t t, uint32_t rt) -> uint32_t {      // device row of tile-row rt of local tile t
        if (SAMPLE) {
            uint32_t j = (tile_first + t * tile_step) * TR + rt;
            if (j >= p.n_sample) j = p.n_sample - 1;
            return sample_row_of(j);                                   // n_sample = 2^sample_shift <= n_rows
        } else {
            const uint32_t r = r0 + t * TR + rt;
            return r > last_row ? last_row : r;
        }
    };
    auto set_tile_ptrs = [&](uint32_t t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t row = tile_rows_of(t, 32 * w + 8 * j + a_pr);
            if (SAMPLE) sRow[(4 * w + j) * 64 + lane] = row;
            else aptr[j] = rows_b + (size_t)row * ld * 4 + a_chunk[j];
        }
    };
    auto a_piece = [&](int j) -> const char* {
        if (SAMPLE) return rows_b + (size_t)sRow[(4 * w + j) * 64 + lane] * ld * 4 + a_chunk[j];
        return aptr[j];
    };
    // The LDS-DMA is issued from inline asm, not through __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass
    // tracks the builtin as a pending LDS write and, at the loop header of the 3-stage ring, cannot bound how many
    // vector-memory operations followed the fill of the image about to be read -- it then puts a vmcnt(0) in front
    // of that stage's first ds_read, which drains the two-stage DMA pipeline.  All ordering between the DMA and the
    // LDS reads is done by hand here (counted s_waitcnt + s_barrier at the top of each stage); compiler-inserted
    // vmcnt waits for ordinary loads stay correct because not counting these instructions only makes them wait longer.
#define VDB_DMA(GP, IMG, LOFF)                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                     \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
    // rows are read once per launch: non-temporal, so that they do not push the queries out of the L2
#define VDB_DMA_NT(GP, IMG, LOFF)                                                                      \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt"                  \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
#define VDB_ISSUE(IMG, KSI)                                                                            \
    {                                                                                                  \
        const uint32_t la_ = (4 * w) * 1024;                                                           \
        const uint32_t lb_ = A_BYTES + (2 * w) * 1024;                                                 \
        const uint32_t ka_ = (KSI) * (KSTAGE * 4);                                                     \
        const uint32_t kb_ = (KSI) * B_BYTES;                                                          \
        if (!(p.ablate & 2u)) {                                                                        \
        const char *g0_ = a_piece(0) + ka_, *g1_ = a_piece(1) + ka_, *g2_ = a_piece(2) + ka_, *g3_ = a_piece(3) + ka_; \
        VDB_DMA_NT(g0_, IMG, la_);                                                                     \
        VDB_DMA_NT(g1_, IMG, la_ + 1024);                                                              \
        VDB_DMA_NT(g2_, IMG, la_ + 2048);                                                              \
        VDB_DMA_NT(g3_, IMG, la_ + 3072);                                                              \
        }                                                                                              \
        if (!(p.ablate & 4u)) {                                                                        \
        VDB_DMA(bbase + (ob[0] + kb_), IMG, lb_);                                                      \
        VDB_DMA(bbase + (ob[1] + kb_), IMG, lb_ + 1024);                                               \
        }                                                                                              \
    }

    // ---- row constants of a tile, one tile ahead, by LDS-DMA (4 bytes per lane): waves 0-3 fetch alpha and the mask
    // word of rows 64(w&3)..+63, waves 4-7 fetch beta.  Issued BEFORE the stage's row/query pieces, so the counted
    // wait at the top of the next stage covers them.
#define VDB_DMA4(GP, LP)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"                        \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)(LP)), "v"((const void*)(GP)) : "memory", "m0")
    auto issue_consts = [&](uint32_t t) {
        const uint32_t par = t & 1u;
        const uint32_t cr = 64 * (w & 3);                              // first tile-row of this wave's chunk
        const uint32_t row = tile_rows_of(t, cr + lane);
        if (w < 4) {
            VDB_DMA4(p.alpha + row, sAlpha + par * TR + cr);
            VDB_DMA4(p.rowmask + (row >> 5), sMaskW + par * TR + cr);
        } else {
            VDB_DMA4(p.beta + row, sBeta + par * TR + cr);
            if (MARGIN) VDB_DMA4(p.margin + row, sMarg + par * TR + cr);
        }
    };
    if (MARGIN) {
        if (tid < TQ) { float g = p.qg[tid]; asm volatile("" : "+v"(g)); sG[tid] = g; }
    }

    f32x16 acc[MT][QT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < QT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets (bytes inside an image)
    const uint32_t swa = (c >> 1) & 7, swb = (c >> 2) & 3;
    const uint32_t a_row_off = (wr * 128 + c) * A_ROWB;                 // + i*32*A_ROWB
    const uint32_t b_row_off = A_BYTES + (wq * 64 + c) * B_ROWB;        // + j*32*B_ROWB
    uint32_t ra[2][2], rb[2];                                           // [k-step][half chunk]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ra[t][0] = ((4 * t + 2 * h) ^ swa) * 16;
        ra[t][1] = ((4 * t + 2 * h + 1) ^ swa) * 16;
        rb[t] = ((2 * t + h) ^ swb) * 16;
    }

    // ---- prologue: constants of tile 0 and stages 0 and 1 in flight
    uint32_t tile = 0, ks = 0;                                          // of the stage being computed
    uint32_t ftile = 0, fks = 0;                                        // of the next stage to fetch
    set_tile_ptrs(0);
    issue_consts(0);
    VDB_ISSUE(sImg0, 0u)
    fks = 1;
    if (fks == KS) { fks = 0; ftile = 1; if (ftile < ntiles) set_tile_ptrs(ftile); }
    if (total > 1) {
        VDB_ISSUE(sImg1, fks)
        ++fks;
        if (fks == KS) { fks = 0; ++ftile; if (ftile < ntiles) set_tile_ptrs(ftile); }
    }

    // STEADY: the caller guarantees st + 2 < total, so the wait and the DMA issue are unconditional.  That is not a
    // micro-optimisation: with a conditional issue hipcc's waitcnt pass sees a path on which nothing follows the
    // previous fill of the image about to be read and puts a vmcnt(0) in front of the first ds_read of every
    // third stage, which drains the two-stage DMA pipeline.
    auto run_stage = [&](uint32_t st, auto buf_tag, auto steady_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* img = BUF == 0 ? sImg0 : BUF == 1 ? sImg1 : sImg2;
        char* img_fill = BUF == 0 ? sImg2 : BUF == 1 ? sImg0 : sImg1;   // stage st+2 goes where stage st-1 was
        // publish stage st: this wave's pieces have landed once at most the 6 pieces of stage st+1 are outstanding
        // (a bare s_barrier: __syncthreads() carries a fence that hipcc lowers to vmcnt(0), which would drain the
        // DMA pipeline at every stage; LDS writes are waited for explicitly, and the asm memory clobbers keep
        // the compiler from moving LDS accesses across)
        if (STEADY || st + 1 < total) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // constants of the NEXT tile into the other parity (every wave is past the epilogue that read it)
        if (ks == 0 && tile + 1 < ntiles) issue_consts(tile + 1);
        if (STEADY || st + 2 < total) {
            VDB_ISSUE(img_fill, fks)
            ++fks;
            if (fks == KS) { fks = 0; ++ftile; if (ftile < ntiles) set_tile_ptrs(ftile); }
        }
        // ---- 2 k-steps of 16: fragments -> bf16 -> 8 MFMAs each
        const char* ap = img + a_row_off;
        const char* bp = img + b_row_off;
        if (!(p.ablate & 1u))
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 fa[MT], fb[QT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float4 lo = *reinterpret_cast<const float4*>(ap + i * 32 * A_ROWB + ra[t][0]);
                const float4 hi = *reinterpret_cast<const float4*>(ap + i * 32 * A_ROWB + ra[t][1]);
                fa[i] = cvt8(lo, hi);
            }
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(bp + j * 32 * B_ROWB + rb[t]);
                fb[j] = __builtin_bit_cast(bf16x8, raw);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < QT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }

        if (ks == KS - 1 && !(p.ablate & 8u)) {
            const uint32_t par = tile & 1u;
            // (the constants of this tile were issued at least one counted top-of-stage wait + barrier ago: every
            // stage that issues them either issues 6 row/query pieces after them or is followed by a vmcnt(0) wait)
            uint32_t tr0;                                               // device row of tile-row 0 (filter mode)
            uint32_t sj0 = 0;                                           // sample index of tile-row 0 (sample mode)
            if (SAMPLE) { sj0 = (tile_first + tile * tile_step) * TR; tr0 = 0; }
            else tr0 = r0 + tile * TR;
            // eligibility of this wave's 128 rows: two ballots over (in range) & (mask bit of the row)
            unsigned long long val[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint32_t rt = wr * 128 + 64 * m + lane;
                bool in;
                uint32_t bit;
                if (SAMPLE) {
                    const uint32_t sj = sj0 + rt;
                    in = sj < p.n_sample;
                    const uint32_t row = sample_row_of(sj);
                    bit = row & 31;
                } else {
                    in = tr0 + rt < r1;
                    bit = rt & 31;                                      // tr0 is a multiple of 32
                }
                val[m] = __ballot(in && ((sMaskW[par * TR + rt] >> bit) & 1u));
            }
            float best_a = __uint_as_float(0x7f800000u), best_b = best_a;   // sample mode: running group minima
            const float* al = sAlpha + par * TR + wr * 128 + 4 * h;
            const float* be = sBeta + par * TR + wr * 128 + 4 * h;
            const float* mg = sMarg + (MARGIN ? par * TR + wr * 128 + 4 * h : 0);
            float ng_a = 0.f, ng_b = 0.f;
            if (MARGIN) { ng_a = -sG[q_a]; ng_b = -sG[q_b]; }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t vbits = (uint32_t)(val[i >> 1] >> (32 * (i & 1) + 4 * h));
                const uint32_t rowb = wr * 128 + i * 32 + 4 * h;       // tile-row of element (j = 0, e = 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 a4 = *reinterpret_cast<const float4*>(al + i * 32 + 8 * j);
                    const float4 b4 = *reinterpret_cast<const float4*>(be + i * 32 + 8 * j);
                    // scores of 4 rows x 2 queries
                    float sa0 = fmaf(acc[i][0][4 * j + 0], a4.x, b4.x), sa1 = fmaf(acc[i][0][4 * j + 1], a4.y, b4.y);
                    float sa2 = fmaf(acc[i][0][4 * j + 2], a4.z, b4.z), sa3 = fmaf(acc[i][0][4 * j + 3], a4.w, b4.w);
                    float sb0 = fmaf(acc[i][1][4 * j + 0], a4.x, b4.x), sb1 = fmaf(acc[i][1][4 * j + 1], a4.y, b4.y);
                    float sb2 = fmaf(acc[i][1][4 * j + 2], a4.z, b4.z), sb3 = fmaf(acc[i][1][4 * j + 3], a4.w, b4.w);
                    if (MARGIN) {
                        const float4 m4 = *reinterpret_cast<const float4*>(mg + i * 32 + 8 * j);
                        sa0 = fmaf(ng_a, m4.x, sa0); sa1 = fmaf(ng_a, m4.y, sa1); sa2 = fmaf(ng_a, m4.z, sa2); sa3 = fmaf(ng_a, m4.w, sa3);
                        sb0 = fmaf(ng_b, m4.x, sb0); sb1 = fmaf(ng_b, m4.y, sb1); sb2 = fmaf(ng_b, m4.z, sb2); sb3 = fmaf(ng_b, m4.w, sb3);
                    }
                    const uint32_t rt0 = rowb + 8 * j;                  // tile-row of element 0
                    if (SAMPLE) {
                        // smallest score of the lane's eligible rows (v_min_f32 skips a NaN score: such a row is no witness
                        // for a threshold, and it reaches the re-rank through the filter pass, which keeps NaN scores)
                        const float inf_ = __uint_as_float(0x7f800000u);
#define VDB_MIN(E, SA, SB)                                                                             \
    {                                                                                                  \
        const bool ok_ = (vbits >> (8 * j + (E))) & 1u;                                                \
        best_a = fminf(best_a, ok_ ? (SA) : inf_);                                                     \
        best_b = fminf(best_b, ok_ ? (SB) : inf_);                                                     \
    }
                        VDB_MIN(0, sa0, sb0) VDB_MIN(1, sa1, sb1) VDB_MIN(2, sa2, sb2) VDB_MIN(3, sa3, sb3)
#undef VDB_MIN
                    } else {
                        // Hits are rare (about 0.2 % of the elements).  Common path per query: four compares whose
                        // lane masks are OR-ed on the scalar unit and ONE not-taken branch; the append code is out of
                        // line.  `!(s > thr)` keeps a NaN score (it must reach the re-rank, flat_index.rs:62).
                        const unsigned long long ma = __builtin_amdgcn_ballot_w64(!(sa0 > thr_a)) | __builtin_amdgcn_ballot_w64(!(sa1 > thr_a)) |
                                                      __builtin_amdgcn_ballot_w64(!(sa2 > thr_a)) | __builtin_amdgcn_ballot_w64(!(sa3 > thr_a));
                        const unsigned long long mb = __builtin_amdgcn_ballot_w64(!(sb0 > thr_b)) | __builtin_amdgcn_ballot_w64(!(sb1 > thr_b)) |
                                                      __builtin_amdgcn_ballot_w64(!(sb2 > thr_b)) | __builtin_amdgcn_ballot_w64(!(sb3 > thr_b));
#define VDB_PUSH(E, S, THR, POOL, PCNT)                                                                \
    if (!((S) > (THR)) && ((vbits >> (8 * j + (E))) & 1u)) {                                           \
        if (PCNT < p.capl) POOL[PCNT] = make_raw_key((S), tr0 + rt0 + (E));                            \
        ++PCNT;                                                                                        \
    }
                        if (__builtin_expect(ma != 0ull, 0)) {
                            VDB_PUSH(0, sa0, thr_a, pool_a, pcnt_a) VDB_PUSH(1, sa1, thr_a, pool_a, pcnt_a)
                            VDB_PUSH(2, sa2, thr_a, pool_a, pcnt_a) VDB_PUSH(3, sa3, thr_a, pool_a, pcnt_a)
                        }
                        if (__builtin_expect(mb != 0ull, 0)) {
                            VDB_PUSH(0, sb0, thr_b, pool_b, pcnt_b) VDB_PUSH(1, sb1, thr_b, pool_b, pcnt_b)
                            VDB_PUSH(2, sb2, thr_b, pool_b, pcnt_b) VDB_PUSH(3, sb3, thr_b, pool_b, pcnt_b)
                        }
#undef VDB_PUSH
                    }
                }
            }
            if (SAMPLE) {
                // one group minimum per (tile, row half, lane half) and query
                const uint32_t g = (((tile_first + tile * tile_step) * 2 + wr) * 2 + h);
                // the key's low word only has to make the keys of one query distinct: the group index
                p.minkeys[(size_t)q_a * p.minkey_stride + g] = best_a < __uint_as_float(0x7f800000u) ? make_key(best_a, g) : EMPTY_KEY;
                p.minkeys[(size_t)q_b * p.minkey_stride + g] = best_b < __uint_as_float(0x7f800000u) ? make_key(best_b, g) : EMPTY_KEY;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < QT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        }
        ++ks;
        if (ks == KS) { ks = 0; ++tile; }
    };

    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    uint32_t st = 0;
    for (; st + 4 < total; st += 3) {                                   // stage index mod 3 == image index
        run_stage(st, B0{}, std::true_type{});
        run_stage(st + 1, B1{}, std::true_type{});
        run_stage(st + 2, B2{}, std::true_type{});
    }
    // the last one to four stages: conditional issue
    if (st < total) { run_stage(st, B0{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B1{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B2{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B0{}, std::false_type{}); ++st; }
    if (!SAMPLE) {
        p.pool_cnt[sub_a] = pcnt_a;
        p.pool_cnt[sub_b] = pcnt_b;
    }
#undef VDB_DMA
#undef VDB_DMA_NT
#undef VDB_DMA4
#undef VDB_ISSUE
}

uint32_t fused_bf16_tile_rows() { return TR; }
uint32_t fused_bf16_subpools_per_query(uint32_t n_wg) { return 4u * n_wg; }
uint32_t fused_bf16_sample_groups(uint32_t n_sample) { return 4u * ((n_sample + TR - 1) / TR); }

#ifdef VDB_DIAG
void launch_fused_bf16(const FusedBf16Params& p, hipStream_t s) {
    if (p.margin) hipLaunchKernelGGL((fused_bf16_kernel<false, true>), dim3(p.n_wg), dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((fused_bf16_kernel<false, false>), dim3(p.n_wg), dim3(NT), 0, s, p);
}
#endif
void launch_sample_bf16(const FusedBf16Params& p, uint32_t n_cu, hipStream_t s) {
    const uint32_t stiles = (p.n_sample + TR - 1) / TR;
    if (!stiles) return;
    (void)n_cu;
    // The sample ALWAYS ranks by the plain score, also when the filter pass ranks by lower-bound scores (p.margin): its
    // instance with the margin fma spills (scratch reloads drain the DMA pipeline), and it does not need it -- any
    // threshold is valid, and the select shifts this one by g_q * (smallest row margin of the index), after which every
    // sample row that met the plain threshold meets the shifted one in lower-bound units (SelectParams::shift_g).
    hipLaunchKernelGGL((fused_bf16_kernel<true, false>), dim3(stiles), dim3(NT), 0, s, p);
}

}  // namespace vdb
