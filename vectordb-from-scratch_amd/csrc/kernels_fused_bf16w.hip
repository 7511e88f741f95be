// kernels_fused_bf16w.hip -- the WIDE filter pass of the screening tier: 128 rows x 512 queries per workgroup.
//
// Same method, same stage loop and the same per-wave work as kernels_fused_bf16p.hip (which documents them): eight waves, each
// 128 rows x 64 queries = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16, f32 rows rounded to bf16 in registers, a 3-image
// LDS-DMA ring with one mid-stage barrier per K stage, the threshold filter in the per-tile epilogue.  What changes is the
// SHAPE of the workgroup's tile: the eight waves sit side by side along the QUERY axis (wave w owns queries 64w .. 64w+63 of
// 512) and all of them read the same 128 rows, where the 256 x 256 kernel has two row halves x four query quarters.  A stage
// image is 16 KB of rows + 32 KB of queries (there: 32 + 16), so a fetched row tile serves 512 queries and a batch of B > 256
// queries reads the database ceil(B / 512) times instead of ceil(B / 256) times -- BASELINE config 3 (B = 1024, SURVEY 8(d):
// "database read once per batch" as the algorithmic traffic) goes from four passes over its 3.84 GB shard to two.  Per stage
// the MFMA work is the same (65536 outputs x 32 K) on half the HBM bytes: the pass is no longer HBM-bound but sits between the
// HBM and the MFMA / LDS-fill rates.
//
// The 512 queries are two consecutive 256-query blocks of the search (query_prep's image layout is [block][K stage][256 x 64 B],
// so a wave's query piece is still 1 KB of contiguous memory); their candidate pools are the two blocks' ordinary
// workgroup-major pools (FusedBf16Params::pool_block_stride), of which this kernel fills sub-pools 0 and 1 (lane halves) and
// zeroes the counts of 2 and 3, so the select's gather, the re-rank and everything downstream are the 256-query code unchanged.
// Scores are bit-identical to the other filter kernels' (same operands, same MFMA order per accumulator).
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
constexpr int TR = 128;                          // rows per tile
constexpr int TQ = 512;                          // queries per tile
constexpr int QB = 256;                          // queries per block of the search (pools, query image)
constexpr int A_ROWB = 128;                      // 32 f32 per row and stage
constexpr int B_ROWB = 64;                       // 32 bf16 per query and stage
constexpr int A_BYTES = TR * A_ROWB;             // 16 KB
constexpr int B_BYTES = TQ * B_ROWB;             // 32 KB
constexpr int BQ_BYTES = QB * B_ROWB;            // 16 KB: one block's query image of a K stage in global memory
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // 48 KB
constexpr int MT = 2, QT = 4;                    // MFMA tiles per wave: 2 x 32 rows, 4 x 32 queries

#ifdef VDB_DIAG
constexpr bool kDiag = true;                     // ablate bits 32 / 64 below exist in the diagnostics build only
#else
constexpr bool kDiag = false;
#endif
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

// MARGIN (Dot / Euclid): the kernel ranks by the LOWER-BOUND score fma(-g_q, margin_row, score) -- see FusedBf16Params.
template <bool MARGIN>
__global__ __launch_bounds__(NT, 2) void fused_bf16w_kernel(FusedBf16Params p) {

    // three DISTINCT LDS objects, each access names its image at compile time (see kernels_fused_dma3.hip)
    __shared__ __attribute__((aligned(16))) char sImg0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg1[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg2[STAGE_BYTES];
    // per-row constants of a tile (alpha, beta, the row's eligibility-mask word), double buffered by tile parity;
    // filled by LDS-DMA as well, so that no wave ever holds a pending ordinary load inside the stage loop
    __shared__ __attribute__((aligned(16))) float sAlpha[2 * TR];
    __shared__ __attribute__((aligned(16))) float sBeta[2 * TR];
    __shared__ __attribute__((aligned(16))) uint32_t sMaskW[2 * TR];
    __shared__ __attribute__((aligned(16))) float sMarg[MARGIN ? 2 * TR : 4];   // per-row error margin of a tile (same double buffering)
    __shared__ float sG[MARGIN ? TQ : 1];                                        // g_q of the 512 queries (read in the epilogue only)
    __shared__ float sThr[TQ];                                                   // their thresholds (likewise)

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile = 64 rows x 128 queries: per K stage a wave reads 8 KB of row fragments + 8 KB of query fragments from LDS -- the
    // split that minimises fragment bytes for 8192 outputs when a row costs 128 B (f32) and a query 64 B (bf16); 128 rows x 64
    // queries (round 3's first version) read 16 + 4 KB and converted twice as many row fragments to bf16
    const uint32_t wr = w >> 2, wq = w & 3;                             // row half of the 128-row tile, query quarter of the 512 queries
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t ld = p.ld;
    const uint32_t KS = ld / KSTAGE;

    // ---- the rows this workgroup covers
    // (sample mode: exactly ONE tile per workgroup, grid = number of sample tiles.  A compile-time tile count lets
    // the compiler drop the next-tile address state; with it the sample instance spilled registers to scratch, and
    // every scratch reload put a vmcnt(0) -- a full drain of the DMA pipeline -- into the stage loop)
    // WHOLE tiles, dealt round-robin: workgroup w takes tiles w, w + n_wg, ...  (see kernels_fused_bf16p.hip)
    const uint32_t nblk = (p.n_rows + TR - 1) / TR;
    const uint32_t r0 = blockIdx.x * TR, r1 = p.n_rows;
    const uint32_t ntiles = blockIdx.x < nblk ? (nblk - blockIdx.x + p.n_wg - 1) / p.n_wg : 0;
    const uint32_t TS = p.n_wg * TR;                                    // rows between consecutive tiles of this workgroup
    // queries of this lane: one column in each of the wave's two 32-query MFMA tiles
    const uint32_t q_a = wq * 128 + c;                                  // the lane's queries: q_a + 32 j, j = 0..3 (one column in each of the wave's four 32-query MFMA tiles)
    // (query j of the lane = query a + 32 j: its count sits 128 j counts and its pool 128 j sub-pools further -- derived where they
    // are needed, in the rare path and at the end, instead of being held in registers across the stage loop: the kernel is at the
    // 256-VGPR limit, and a pointer spilled to scratch cost a vmcnt(0) -- a drain of the DMA ring -- at every append; the thresholds
    // live in LDS for the same reason)
    // sub-pool r = 2 * row half + lane half of query q (inside its 256-query block), as in kernels_fused_bf16p.hip; recomputed from
    // the thread index wherever it is needed (the empty asm keeps hipcc from hoisting the result into a register held -- i.e.
    // spilled -- across the stage loop)
    auto sub_of = [&](uint32_t jq) -> size_t {
        uint32_t t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
        const uint32_t l_ = t_ & 63u, w_ = t_ >> 6;
        const uint32_t q_ = (w_ & 3u) * 128u + (l_ & 31u) + 32u * jq;
        return ((size_t)blockIdx.x * QB + (q_ & 255u)) * 4 + 2 * (w_ >> 2) + (l_ >> 5);
    };
    // the lane's four sub-pool counts, two 16-bit counters per register (saturating; a count above capl means overflow): four
    // registers' worth of counters were the ones the MARGIN instance spilled, and a scratch access on the append path drains the DMA ring
    uint32_t pcnt_pk[2] = {0u, 0u};
    // can a score of this launch be NaN at all?  (wave-uniform; decides how the epilogue tests four scores at once)
    const bool no_nan = fused_no_nan(p.scalars, p.qmax_bits, !MARGIN);
    if (ntiles == 0) {
#pragma unroll
        for (int j = 0; j < QT; ++j) p.pool_cnt[(size_t)(wq >> 1) * p.cnt_block_stride + sub_of(j)] = 0;
        return;
    }
    const uint32_t total = ntiles * KS;
    const uint32_t last_row = p.n_rows - 1;
    const char* __restrict__ rows_b = reinterpret_cast<const char*>(p.rows);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qb);

    // ---- DMA plan.  A stage image = 16 row pieces + 32 query pieces of 1 KB.  Wave w fills row pieces 2w, 2w+1 (8 rows x
    // 128 B each: lane L -> row L>>3, 16-byte position L&7) and query pieces 4w .. 4w+3 (16 queries x 64 B each).  Both images
    // are XOR-swizzled so that the fragment reads below are bank-conflict free: data chunk x of row r sits at position
    // x ^ ((r>>1)&7), data chunk x of query r at position x ^ ((r>>2)&3); the filling lane fetches the permuted source chunk.
    const uint32_t a_pr = lane >> 3, a_pp = lane & 7;
    // tile-row of piece j: rt = 16w + 8j + a_pr, so (rt>>1)&7 = (4j + (a_pr>>1)) & 7: one source chunk for j = 0, one for j = 1
    const uint32_t a_chunk0 = (a_pp ^ ((a_pr >> 1) & 7)) * 16, a_chunk1 = (a_pp ^ ((4 + (a_pr >> 1)) & 7)) * 16;
    // (the queries are stored by query_prep in image order, one 16 KB image per 256-query block and K stage: the 32 KB image of a
    // stage is block 0's image followed by block 1's, and a wave's piece is 1 KB of CONTIGUOUS global memory; waves 0-3 fill
    // block 0's half, waves 4-7 block 1's)
    // ONE per-lane offset for the wave's four query pieces (they are 1 KB apart: the instruction's immediate offset does the
    // rest) -- this kernel sits at the 256-VGPR limit like its sibling.  32 bits suffice: ld <= 16384 puts a block at 8 MB.
    const uint32_t ob0 = (w >> 2) * (uint32_t)(QB * 2) * ld + (4 * (w & 3)) * 1024 + lane * 16;
    // the tile's rows are contiguous (the store is allocated and zero-filled in multiples of 256 rows, so tile rows past the
    // last row are readable; the eligibility ballots of the epilogue keep them out)
    const char* aptr0 = nullptr; const char* aptr1 = nullptr;
    auto tile_rows_of = [&](uint32_t t, uint32_t rt) -> uint32_t {      // device row of tile-row rt of local tile t
        const uint32_t r = r0 + t * TS + rt;
        return r > last_row ? last_row : r;
    };
    auto set_tile_ptrs = [&](uint32_t t) {
        const uint32_t row = r0 + t * TS + 16 * w + a_pr;              // unclamped, see above
        aptr0 = rows_b + (size_t)row * ld * 4 + a_chunk0;
        aptr1 = rows_b + (size_t)(row + 8) * ld * 4 + a_chunk1;
    };
    auto a_piece = [&](int j) -> const char* { return j ? aptr1 : aptr0; };
    // The LDS-DMA is issued from inline asm, not through __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass
    // tracks the builtin as a pending LDS write and, at the loop header of the 3-stage ring, cannot bound how many
    // vector-memory operations followed the fill of the image about to be read -- it then puts a vmcnt(0) in front
    // of that stage's first ds_read, which drains the two-stage DMA pipeline.  All ordering between the DMA and the
    // LDS reads is done by hand here (counted s_waitcnt + s_barrier at the top of each stage); compiler-inserted
    // vmcnt waits for ordinary loads stay correct because not counting these instructions only makes them wait longer.
#define VDB_DMA(GP, IMG, LOFF)                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                     \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
    // (the instruction's immediate offset applies to BOTH addresses -- the global source and the LDS destination M0 + offset +
    // lane * 16 -- so the four query pieces of a wave share one source register AND one M0 value)
#define VDB_DMA_OFF(GP, IMG, LOFF, IMM)                                                                 \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%2"           \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)), "i"(IMM) : "memory", "m0")
    // rows are read once per launch: non-temporal, so that they do not push the queries out of the L2
#define VDB_DMA_NT(GP, IMG, LOFF)                                                                      \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt"                  \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
#define VDB_ISSUE(IMG, KSI)                                                                            \
    {                                                                                                  \
        const uint32_t la_ = (2 * w) * 1024;                                                           \
        const uint32_t lb_ = A_BYTES + (4 * w) * 1024;                                                 \
        const uint32_t ka_ = (KSI) * (KSTAGE * 4);                                                     \
        const uint32_t kb_ = (KSI) * BQ_BYTES;                                                         \
        if (!(p.ablate & 2u)) {                                                                        \
        VDB_DMA_NT(a_piece(0) + ka_, IMG, la_);                                                        \
        VDB_DMA_NT(a_piece(1) + ka_, IMG, la_ + 1024);                                                 \
        }                                                                                              \
        if (!(p.ablate & 4u)) {                                                                        \
        VDB_DMA_OFF(bbase + (ob0 + kb_), IMG, lb_, 0);                                                 \
        VDB_DMA_OFF(bbase + (ob0 + kb_), IMG, lb_, 1024);                                              \
        VDB_DMA_OFF(bbase + (ob0 + kb_), IMG, lb_, 2048);                                              \
        VDB_DMA_OFF(bbase + (ob0 + kb_), IMG, lb_, 3072);                                              \
        }                                                                                              \
    }

    // ---- row constants of a tile, one tile ahead, by LDS-DMA (4 bytes per lane): waves 0-1 fetch alpha and the mask word of
    // rows 64(w&1)..+63, waves 2-3 beta (and the margin).  Issued BEFORE the stage's row/query pieces, so the counted wait at
    // the top of the next stage covers them.
#define VDB_DMA4(GP, LP)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"                        \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)(LP)), "v"((const void*)(GP)) : "memory", "m0")
    auto issue_consts = [&](uint32_t t) {
        if (w >= 4) return;
        const uint32_t par = t & 1u;
        const uint32_t cr = 64 * (w & 1);                              // first tile-row of this wave's chunk
        const uint32_t row = tile_rows_of(t, cr + lane);
        if (w < 2) {
            VDB_DMA4(p.alpha + row, sAlpha + par * TR + cr);
            VDB_DMA4(p.rowmask + (row >> 5), sMaskW + par * TR + cr);
        } else {
            VDB_DMA4(p.beta + row, sBeta + par * TR + cr);
            if (MARGIN) VDB_DMA4(p.margin + row, sMarg + par * TR + cr);
        }
    };
    if (MARGIN) {
        // g_q lives in LDS, not in two more registers per lane held across the stage loop (the kernel sits at the 256-VGPR
        // limit); the load is consumed here so that no ordinary load is pending inside the loop, and the prologue's barrier
        // publishes the array
        if (tid < TQ) { float g = p.qg[tid]; asm volatile("" : "+v"(g)); sG[tid] = g; }
    }
    {
        // thresholds into LDS: consumed here, so that no ordinary load is pending inside the stage loop (a first use there would
        // get a compiler-inserted vmcnt(0), i.e. a wait for every DMA in flight, once per tile)
        float t = p.thr[tid];                                           // NT == TQ
        if (p.ablate & 16u) t = -3.0e38f;                               // diagnostics: nothing passes the filter (finite, so that the MARGIN instance's loosened threshold is not inf - inf)
        asm volatile("" : "+v"(t));
        sThr[tid] = t;
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
    const uint32_t a_row_off = (wr * 64 + c) * A_ROWB;                  // + i*32*A_ROWB
    const uint32_t b_row_off = A_BYTES + (wq * 128 + c) * B_ROWB;       // + j*32*B_ROWB
    uint32_t ra[2], rb[2];                                              // [k-step]; the second half chunk of a row fragment is at ra ^ 16
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ra[t] = ((4 * t + 2 * h) ^ swa) * 16;
        rb[t] = ((2 * t + h) ^ swb) * 16;
    }

    // ---- prologue: constants of tile 0 and stages 0, 1, 2 in flight; publish stage 0
    uint32_t tile = 0, ks = 0;                                          // of the stage being computed
    uint32_t ftile = 0, fks = 0;                                        // of the next stage to fetch
#define VDB_ADV { ++fks; if (fks == KS) { fks = 0; ++ftile; if (ftile < ntiles) set_tile_ptrs(ftile); } }
    set_tile_ptrs(0);
    issue_consts(0);
    VDB_ISSUE(sImg0, fks) VDB_ADV
    if (total > 1) { VDB_ISSUE(sImg1, fks) VDB_ADV }
    if (total > 2) { VDB_ISSUE(sImg2, fks) VDB_ADV }
    if (total > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (total > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 fa0[MT], fb0[QT];                                            // k-step 0 fragments of the stage to compute next
#define VDB_LOAD_FRAGS(FA, FB, IMG, T_)                                                                \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                            \
            const float4 lo_ = *reinterpret_cast<const float4*>((IMG) + a_row_off + i_ * 32 * A_ROWB + ra[T_]); \
            const float4 hi_ = *reinterpret_cast<const float4*>((IMG) + a_row_off + i_ * 32 * A_ROWB + (ra[T_] ^ 16u)); \
            FA[i_] = cvt8(lo_, hi_);                                                                   \
        }                                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < QT; ++j_) {                                            \
            const u32x4 raw_ = *reinterpret_cast<const u32x4*>((IMG) + b_row_off + j_ * 32 * B_ROWB + rb[T_]); \
            FB[j_] = __builtin_bit_cast(bf16x8, raw_);                                                 \
        }                                                                                              \
    }
    VDB_LOAD_FRAGS(fa0, fb0, sImg0, 0)

    // STEADY: the caller guarantees st + 3 < total, so the wait and the DMA issue are unconditional.
    auto run_stage = [&](uint32_t st, auto buf_tag, auto steady_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool STEADY = decltype(steady_tag)::value;
        char* img = BUF == 0 ? sImg0 : BUF == 1 ? sImg1 : sImg2;
        const char* nxt = BUF == 0 ? sImg1 : BUF == 1 ? sImg2 : sImg0;
        bf16x8 fa1[MT], fb1[QT];
        // k-step 0 MFMAs with the fragment reads of k-step 1 between them, one row block at a time: the two ds_read_b128 of
        // a row fragment are issued, two MFMAs run, then the fragment is rounded to bf16 -- at most one f32 fragment (8
        // registers) is in flight, not four (reading all of them first spilled)
#define VDB_READ_B(FB, IMG, T_)                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < QT; ++j_) {                                                \
        const u32x4 raw_ = *reinterpret_cast<const u32x4*>((IMG) + b_row_off + j_ * 32 * B_ROWB + rb[T_]); \
        FB[j_] = __builtin_bit_cast(bf16x8, raw_);                                                     \
    }
#define VDB_STEP(I_, FA_USE, FB_USE, FA_NEW, IMG_NEW, T_NEW, LOAD_, EX0, EX1, EX2)                     \
    {                                                                                                  \
        float4 lo_, hi_;                                                                               \
        if (LOAD_) {                                                                                   \
            lo_ = *reinterpret_cast<const float4*>((IMG_NEW) + a_row_off + (I_) * 32 * A_ROWB + ra[T_NEW]); \
            hi_ = *reinterpret_cast<const float4*>((IMG_NEW) + a_row_off + (I_) * 32 * A_ROWB + (ra[T_NEW] ^ 16u)); \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        acc[I_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[0], acc[I_][0], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        EX0                                                                                            \
        acc[I_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[1], acc[I_][1], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        EX1                                                                                            \
        acc[I_][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[2], acc[I_][2], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        EX2                                                                                            \
        acc[I_][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[3], acc[I_][3], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (LOAD_) FA_NEW[I_] = cvt8(lo_, hi_);                                                        \
    }
        VDB_READ_B(fb1, img, 1)
        VDB_STEP(0, fa0, fb0, fa1, img, 1, true, , , ) VDB_STEP(1, fa0, fb0, fa1, img, 1, true, , , )
        // publish stage st+1: this wave's pieces of it have landed once at most the 6 pieces of stage st+2 are
        // outstanding; lgkmcnt(0): this wave's fragment reads of stage st are done, so after the barrier the image of
        // stage st is free.  (A bare s_barrier: __syncthreads() carries a fence that hipcc lowers to vmcnt(0).)
        if (STEADY || st + 2 < total) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // constants of the NEXT tile into the other parity (every wave is past the epilogue that read it)
        if (ks == 0 && tile + 1 < ntiles) issue_consts(tile + 1);
        // k-step 1 MFMAs, between them the fragment reads of k-step 0 of stage st+1 and the six DMA pieces of stage st+3
        // (under a wave-uniform branch; the MFMA sequence itself is one straight line)
        const bool more = STEADY || st + 1 < total;
        const bool do_dma = STEADY || st + 3 < total;
        const uint32_t la_ = (2 * w) * 1024;
        const uint32_t lb_ = A_BYTES + (4 * w) * 1024;
        const uint32_t ka_ = fks * (KSTAGE * 4);
        const uint32_t kb_ = fks * BQ_BYTES;
        const char* const qsrc_ = bbase + (ob0 + kb_);
#define VDB_PIECE_A(J_)                                                                                \
    if (do_dma && !(p.ablate & 2u)) { VDB_DMA_NT(a_piece(J_) + ka_, img, la_ + (J_) * 1024); }         \
    __builtin_amdgcn_sched_barrier(0);
#define VDB_PIECE_B(J_)                                                                                \
    if (do_dma && !(p.ablate & 4u)) { VDB_DMA_OFF(qsrc_, img, lb_, (J_) * 1024); }                     \
    __builtin_amdgcn_sched_barrier(0);
        if (more) VDB_READ_B(fb0, nxt, 0)
        VDB_STEP(0, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_A(0), VDB_PIECE_A(1), VDB_PIECE_B(0))
        VDB_STEP(1, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_B(1), VDB_PIECE_B(2), VDB_PIECE_B(3))
#undef VDB_PIECE_A
#undef VDB_PIECE_B
#undef VDB_STEP
#undef VDB_READ_B
        if (do_dma) VDB_ADV

        if (ks == KS - 1 && !(p.ablate & 8u)) {
            const uint32_t par = tile & 1u;
            // (the constants of this tile were issued at least one counted top-of-stage wait + barrier ago: every
            // stage that issues them either issues 6 row/query pieces after them or is followed by a vmcnt(0) wait)
            const uint32_t tr0 = r0 + tile * TS;                        // device row of tile-row 0
            // eligibility of this wave's 64 rows: one ballot over (in range) & (mask bit of the row)
            const uint32_t rt_l = wr * 64 + lane;
            const unsigned long long val = __ballot(tr0 + rt_l < r1 && ((sMaskW[par * TR + rt_l] >> (rt_l & 31)) & 1u));   // tr0 is a multiple of 32
            const float* al = sAlpha + par * TR + wr * 64 + 4 * h;
            const float* be = sBeta + par * TR + wr * 64 + 4 * h;
            // MARGIN: the filter is  lb = fma(-g_q, margin_row, score) <= thr.  Since margin_row <= mmax (the largest margin of
            // this wave's 64 rows), lb <= thr implies score <= thr + g_q mmax =: thp -- so the COMMON path compares the plain
            // score with a per-tile loosened threshold, and only the rare path computes lb and applies the exact test.  The
            // slack covers the f32 rounding of thp and of lb, so no row with lb <= thr can fail the pre-test.
            const float* mg = sMarg + (MARGIN ? par * TR + wr * 64 : 0);
            float thr_q[QT], thp[QT], ng[QT];
#pragma unroll
            for (int jq = 0; jq < QT; ++jq) { thr_q[jq] = sThr[q_a + 32 * jq]; thp[jq] = thr_q[jq]; ng[jq] = 0.f; }
            if (MARGIN) {
                float mm = mg[lane];                                    // +inf margins (norm overflow) open the tile; NaN rows carry NaN scores anyway
                for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
#pragma unroll
                for (int jq = 0; jq < QT; ++jq) {
                    const float g = sG[q_a + 32 * jq];
                    ng[jq] = -g;
                    thp[jq] = fmaf(g, mm, thr_q[jq]); thp[jq] += (fabsf(thr_q[jq]) + g * mm) * 6.0e-7f;
                }
            }
            // Per row block i (32 rows x 4 queries per lane = 4 groups of 4 rows): the COMMON path is branch-free -- scores of the
            // four groups, the minimum of each group against the (loosened) threshold, the lane masks OR-ed on the scalar unit -- and
            // ends in ONE not-taken branch per query; one pair of LDS reads (alpha, beta of 4 rows) serves all four queries.  Hits
            // are rare (about 0.1 % of the elements), so the RARE path recomputes the block's scores for its query from the
            // accumulators -- nothing of the common path has to stay live for it -- and appends what passes the exact test.
            // v_min_f32 drops a NaN operand and a NaN score must pass (flat_index.rs:62): the minimum test stands alone only when
            // no score of the launch can be NaN (fused_no_nan: every norm within [2^-40, 2^40]); otherwise a NaN-propagating sum of
            // each group is tested as well (inf - inf gives a false alarm, which the exact per-row test of the rare path sorts out).
#define VDB_SCORES(I_, J_, Q_, S01, S23)                                                               \
    const float4 a4_ = *reinterpret_cast<const float4*>(al + (I_) * 32 + 8 * (J_));                    \
    const float4 b4_ = *reinterpret_cast<const float4*>(be + (I_) * 32 + 8 * (J_));                    \
    const f32x2 al01_ = {a4_.x, a4_.y}, al23_ = {a4_.z, a4_.w}, be01_ = {b4_.x, b4_.y}, be23_ = {b4_.z, b4_.w}; \
    const f32x2 p01_ = {acc[I_][Q_][4 * (J_) + 0], acc[I_][Q_][4 * (J_) + 1]}, p23_ = {acc[I_][Q_][4 * (J_) + 2], acc[I_][Q_][4 * (J_) + 3]}; \
    const f32x2 S01 = __builtin_elementwise_fma(p01_, al01_, be01_), S23 = __builtin_elementwise_fma(p23_, al23_, be23_);
            // the append: one 4-bit hit mask per lane and group, then a loop over its set bits -- typically one lane, one iteration
#define VDB_APPEND(I_, J_, S0, S1, S2, S3, THP, THR, NG, POOL, PCNT)                                   \
    {                                                                                                  \
        uint32_t hm_ = (!((S0) > (THP)) ? 1u : 0u) | (!((S1) > (THP)) ? 2u : 0u) | (!((S2) > (THP)) ? 4u : 0u) | (!((S3) > (THP)) ? 8u : 0u); \
        hm_ &= (vbits >> (8 * (J_))) & 0xfu;                                                           \
        while (hm_) {                                                                                  \
            const uint32_t e_ = (uint32_t)__builtin_ctz(hm_);                                          \
            hm_ &= hm_ - 1u;                                                                           \
            float sc_ = e_ == 0 ? (S0) : e_ == 1 ? (S1) : e_ == 2 ? (S2) : (S3);                       \
            if (MARGIN) {                                              /* the exact test, on the lower-bound score */ \
                sc_ = fmaf((NG), mg[(I_) * 32 + 8 * (J_) + 4 * h + e_], sc_);                          \
                if (sc_ > (THR)) continue;                                                             \
            }                                                                                          \
            if (!(kDiag && (p.ablate & 32u)) && PCNT < p.capl) POOL[PCNT] = make_raw_key(sc_, tr0 + rowb_ + 8 * (J_) + e_); /* diag 32: count only */ \
            ++PCNT;                                                                                    \
        }                                                                                              \
    }
#define VDB_RARE(I_, Q_)                                                                               \
    {                                                                                                  \
        uint64_t* const pool_q = p.pool + (size_t)(wq >> 1) * p.pool_block_stride + sub_of(Q_) * p.capl;   /* queries 128 wq .. + 127: block wq >> 1 */ \
        uint32_t hh_ = h;                      /* tile-row of element (j = 0, e = 0), recomputed here for the same reason */ \
        asm volatile("" : "+v"(hh_));                                                                  \
        const uint32_t rowb_ = wr * 64 + (I_) * 32 + 4 * hh_;                                          \
        uint32_t cnt_ = (pcnt_pk[(Q_) >> 1] >> (16 * ((Q_) & 1))) & 0xffffu;                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                             \
            if (hit[Q_][j_] != 0ull) {         /* only the groups of four rows in which some lane has a hit (wave-uniform) */ \
                VDB_SCORES(I_, j_, Q_, r01_, r23_)                                                     \
                VDB_APPEND(I_, j_, r01_.x, r01_.y, r23_.x, r23_.y, thp[Q_], thr_q[Q_], ng[Q_], pool_q, cnt_) \
            }                                                                                          \
        }                                                                                              \
        if (cnt_ > 0xffffu) cnt_ = 0xffffu;                                                            \
        pcnt_pk[(Q_) >> 1] = (pcnt_pk[(Q_) >> 1] & ~(0xffffu << (16 * ((Q_) & 1)))) | (cnt_ << (16 * ((Q_) & 1))); \
    }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t vbits = (uint32_t)(val >> (32 * i + 4 * h));
                // lanes with a possible hit, per query and per group of four rows: at k = 100 (config 3) two of three 32 x 32 blocks hold a
                // hit somewhere, but only one or two of a block's four groups do -- the rare path recomputes just those
                unsigned long long hit[QT][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // scores of 4 rows x 4 queries (two rows per v_pk_fma_f32: the same IEEE fma per element, half the instructions)
                    const float4 a4 = *reinterpret_cast<const float4*>(al + i * 32 + 8 * j);
                    const float4 b4 = *reinterpret_cast<const float4*>(be + i * 32 + 8 * j);
                    const f32x2 al01 = {a4.x, a4.y}, al23 = {a4.z, a4.w}, be01 = {b4.x, b4.y}, be23 = {b4.z, b4.w};
#pragma unroll
                    for (int jq = 0; jq < QT; ++jq) {
                        const f32x2 p01 = {acc[i][jq][4 * j + 0], acc[i][jq][4 * j + 1]}, p23 = {acc[i][jq][4 * j + 2], acc[i][jq][4 * j + 3]};
                        const f32x2 r01 = __builtin_elementwise_fma(p01, al01, be01), r23 = __builtin_elementwise_fma(p23, al23, be23);
                        const f32x2 n_ = __builtin_elementwise_min(r01, r23);
                        hit[jq][j] = __builtin_amdgcn_ballot_w64(!(fminf(n_.x, n_.y) > thp[jq]));
                    }
                    // two groups in flight at a time: with all four the register allocator spills (the kernel sits at 256 VGPRs,
                    // and a scratch access in here costs a vmcnt(0), i.e. a drain of the DMA ring, per tile)
                    if (j & 1) __builtin_amdgcn_sched_barrier(0);
                }
                if (__builtin_expect(!no_nan, 0)) {                     // a real (wave-uniform) branch: the empty asm keeps hipcc from
                    asm volatile("" ::: "memory");                      // computing the sums always and selecting with v_cndmask
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        { VDB_SCORES(i, j, 0, u01, u23) const f32x2 u_ = u01 + u23; const float t_ = u_.x + u_.y; hit[0][j] |= __builtin_amdgcn_ballot_w64(t_ != t_); }
                        { VDB_SCORES(i, j, 1, u01, u23) const f32x2 u_ = u01 + u23; const float t_ = u_.x + u_.y; hit[1][j] |= __builtin_amdgcn_ballot_w64(t_ != t_); }
                        { VDB_SCORES(i, j, 2, u01, u23) const f32x2 u_ = u01 + u23; const float t_ = u_.x + u_.y; hit[2][j] |= __builtin_amdgcn_ballot_w64(t_ != t_); }
                        { VDB_SCORES(i, j, 3, u01, u23) const f32x2 u_ = u01 + u23; const float t_ = u_.x + u_.y; hit[3][j] |= __builtin_amdgcn_ballot_w64(t_ != t_); }
                    }
                }
                if (__builtin_expect((hit[0][0] | hit[0][1] | hit[0][2] | hit[0][3]) != 0ull, 0)) VDB_RARE(i, 0)
                if (__builtin_expect((hit[1][0] | hit[1][1] | hit[1][2] | hit[1][3]) != 0ull, 0)) VDB_RARE(i, 1)
                if (__builtin_expect((hit[2][0] | hit[2][1] | hit[2][2] | hit[2][3]) != 0ull, 0)) VDB_RARE(i, 2)
                if (__builtin_expect((hit[3][0] | hit[3][1] | hit[3][2] | hit[3][3]) != 0ull, 0)) VDB_RARE(i, 3)
            }
#undef VDB_RARE
#undef VDB_APPEND
#undef VDB_SCORES
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
    for (; st + 5 < total; st += 3) {                                   // stage index mod 3 == image index
        run_stage(st, B0{}, std::true_type{});
        run_stage(st + 1, B1{}, std::true_type{});
        run_stage(st + 2, B2{}, std::true_type{});
    }
    // the last one to five stages: conditional issue
    if (st < total) { run_stage(st, B0{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B1{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B2{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B0{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, B1{}, std::false_type{}); ++st; }
#pragma unroll
    for (int j = 0; j < QT; ++j) p.pool_cnt[(size_t)(wq >> 1) * p.cnt_block_stride + sub_of(j)] = (pcnt_pk[j >> 1] >> (16 * (j & 1))) & 0xffffu;
#undef VDB_DMA
#undef VDB_DMA_NT
#undef VDB_DMA_OFF
#undef VDB_DMA4
#undef VDB_ISSUE
#undef VDB_ADV
#undef VDB_LOAD_FRAGS
}

uint32_t fused_bf16w_tile_rows() { return TR; }

void launch_fused_bf16w(const FusedBf16Params& p, hipStream_t s) {
    if (p.margin) hipLaunchKernelGGL((fused_bf16w_kernel<true>), dim3(p.n_wg), dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((fused_bf16w_kernel<false>), dim3(p.n_wg), dim3(NT), 0, s, p);
}

}  // namespace vdb
