// kernels_fused_bf16p.hip -- the FILTER pass of the screening tier (kernels_fused_bf16.hip, which keeps the sample pass
// and documents the method), with the stage loop software-pipelined around ONE barrier per stage that sits in the
// MIDDLE of the stage:
//     read the fragments of k-step 1 of stage s (f32 -> bf16 in registers)   (LDS, overlaps the next line)
//     8 MFMAs of k-step 0                                                     (fragments read during stage s-1)
//     wait for this wave's pieces of stage s+1, barrier     -> stage s+1 is published, and nobody reads the image of
//                                                              stage s any more (every wave waited for its LDS reads)
//     read the fragments of k-step 0 of stage s+1
//     8 MFMAs of k-step 1, with the 6 DMA pieces of stage s+3 -- INTO THE IMAGE OF STAGE s -- issued between them
// so the LDS fragment traffic runs under the MFMAs instead of in front of them, the vector-memory issue of eight
// in-phase waves no longer precedes their first MFMA, and the 3-image ring holds the stage being computed plus two in
// flight, refilled half a stage earlier.  Scores are bit-identical to the unpipelined kernel (same operands, same MFMA
// order per accumulator).  VDB_FUSED_PIPE=0 selects the unpipelined filter pass (A/B runs).
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
// One more packed FMA per two elements of the epilogue and one more per-row constant staged per tile; the stage loop is
// untouched.  Cosine runs the MARGIN = false instance (its row error is bounded relative to the row's own norm).
template <bool SAMPLE, bool MARGIN>
__global__ __launch_bounds__(NT, 2) void fused_bf16p_kernel(FusedBf16Params p) {
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
    __shared__ float sG[MARGIN ? TQ : 1];                                        // g_q of the 256 queries (read in the epilogue only)

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
        // WHOLE tiles, dealt round-robin: workgroup w takes tiles w, w + n_wg, ...  (some workgroups run one tile fewer; the
        // last tiles of the others meet an HBM that is no longer contended).  Round-robin rather than one contiguous range per
        // workgroup: the 256 streams then walk through ONE window of the matrix together instead of 256 windows 12 MB apart
        // (tools/read_pattern_probe.hip: 0.465 against 0.471 ms for the DMA traffic of this kernel alone)
        const uint32_t nblk = (p.n_rows + TR - 1) / TR;
        r0 = blockIdx.x * TR;
        r1 = p.n_rows;
        ntiles_rt = blockIdx.x < nblk ? (nblk - blockIdx.x + p.n_wg - 1) / p.n_wg : 0;
    }
    const uint32_t TS = SAMPLE ? TR : p.n_wg * TR;                      // rows between consecutive tiles of this workgroup
    const uint32_t ntiles = SAMPLE ? 1u : ntiles_rt;
    // queries of this lane: one column in each of the wave's two 32-query MFMA tiles
    const uint32_t q_a = wq * 64 + c, q_b = q_a + 32;
    uint64_t* pool_a = nullptr; uint64_t* pool_b = nullptr;
    size_t sub_a = 0, sub_b = 0;
    float thr_a = 0.f, thr_b = 0.f;
    if (!SAMPLE) {
        // the COUNTS are workgroup-major as well (since the end of round 2): the 1024 counts of a workgroup are one 4 KB block
        // written in whole lines, not 1024 four-byte stores 4 KB apart (262 000 scattered stores per launch, ~10 us of tail)
        sub_a = (((size_t)blockIdx.x * TQ + q_a) * 2 + wr) * 2 + h;
        sub_b = (((size_t)blockIdx.x * TQ + q_b) * 2 + wr) * 2 + h;
        // The pool KEYS are laid out workgroup-major -- slot ((wg*256 + q)*4 + row half*2 + lane half)*capl -- so that the
        // few scattered appends of one workgroup fall into ONE 2 MB region instead of one region per query (256 regions
        // 2 MB apart: every append then missed the CU's address-translation cache in front of the row stream).  The
        // select's gather knows both layouts (SelectParams::wg_major).
        pool_a = p.pool + ((((size_t)blockIdx.x * TQ + q_a) * 2 + wr) * 2 + h) * p.capl;
        pool_b = p.pool + ((((size_t)blockIdx.x * TQ + q_b) * 2 + wr) * 2 + h) * p.capl;
        thr_a = p.thr[q_a];
        thr_b = p.thr[q_b];
        // consume the two loads here: a first use inside the stage loop would get a compiler-inserted vmcnt(0)
        // there, i.e. a wait for every DMA in flight, once per tile
        if (p.ablate & 16u) thr_a = thr_b = -3.0e38f;              // diagnostics: nothing passes the filter (cost of the append path; finite, so that the MARGIN instance's loosened threshold is not inf - inf)
        asm volatile("" : "+v"(thr_a), "+v"(thr_b));
    }
    uint32_t pcnt_a = 0, pcnt_b = 0;
    // can a score of this launch be NaN at all?  (wave-uniform; decides how the epilogue tests four scores at once)
    const bool no_nan = !SAMPLE && fused_no_nan(p.scalars, p.qmax_bits, !MARGIN);
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
    // tile-row of piece j: rt = 32w + 8j + a_pr, so (rt>>1)&7 = (4(j&1) + (a_pr>>1)) & 7: one source chunk for even j, one for odd j
    const uint32_t a_chunk0 = (a_pp ^ ((a_pr >> 1) & 7)) * 16, a_chunk1 = (a_pp ^ ((4 + (a_pr >> 1)) & 7)) * 16;
    // (the queries are stored by query_prep in exactly this image order, one 16 KB image per K stage: a wave's
    // query piece is 1 KB of CONTIGUOUS global memory -- 8 full 128-byte requests instead of 16 scattered 64-byte ones)
    const uint32_t ob[2] = {(2 * w) * 1024 + lane * 16, (2 * w + 1) * 1024 + lane * 16};
    (void)b_pr; (void)b_pp;
    // the tile's rows are contiguous (the store is allocated and zero-filled in multiples of 256 rows, so tile rows past
    // the last row are readable; the eligibility ballots of the epilogue keep them out): pieces j and j+2 are 16 rows
    // apart -> two base pointers (even j, odd j) and a uniform stride instead of four 64-bit addresses per lane
    const char* aptr0 = nullptr; const char* aptr1 = nullptr;
    const size_t a_pair_stride = (size_t)16 * ld * 4;
    auto tile_rows_of = [&](uint32_t t, uint32_t rt) -> uint32_t {      // device row of tile-row rt of local tile t
        if (SAMPLE) {
            uint32_t j = (tile_first + t * tile_step) * TR + rt;
            if (j >= p.n_sample) j = p.n_sample - 1;
            return sample_row_of(j);                                   // n_sample = 2^sample_shift <= n_rows
        } else {
            const uint32_t r = r0 + t * TS + rt;
            return r > last_row ? last_row : r;
        }
    };
    auto set_tile_ptrs = [&](uint32_t t) {
        const uint32_t row = r0 + t * TS + 32 * w + a_pr;              // unclamped, see above
        aptr0 = rows_b + (size_t)row * ld * 4 + a_chunk0;
        aptr1 = rows_b + (size_t)(row + 8) * ld * 4 + a_chunk1;
    };
    auto a_piece = [&](int j) -> const char* { return ((j & 1) ? aptr1 : aptr0) + (size_t)(j >> 1) * a_pair_stride; };
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
        VDB_DMA_NT(a_piece(0) + ka_, IMG, la_);                                                        \
        VDB_DMA_NT(a_piece(1) + ka_, IMG, la_ + 1024);                                                 \
        VDB_DMA_NT(a_piece(2) + ka_, IMG, la_ + 2048);                                                 \
        VDB_DMA_NT(a_piece(3) + ka_, IMG, la_ + 3072);                                                 \
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
        // g_q lives in LDS, not in two more registers per lane held across the stage loop (the kernel sits at the 256-VGPR
        // limit); the load is consumed here so that no ordinary load is pending inside the loop, and the prologue's barrier
        // publishes the array
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
#define VDB_STEP(I_, FA_USE, FB_USE, FA_NEW, IMG_NEW, T_NEW, LOAD_, EXTRA)                             \
    {                                                                                                  \
        float4 lo_, hi_;                                                                               \
        if (LOAD_) {                                                                                   \
            lo_ = *reinterpret_cast<const float4*>((IMG_NEW) + a_row_off + (I_) * 32 * A_ROWB + ra[T_NEW]); \
            hi_ = *reinterpret_cast<const float4*>((IMG_NEW) + a_row_off + (I_) * 32 * A_ROWB + (ra[T_NEW] ^ 16u)); \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        acc[I_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[0], acc[I_][0], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        EXTRA                                                                                          \
        acc[I_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA_USE[I_], FB_USE[1], acc[I_][1], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (LOAD_) FA_NEW[I_] = cvt8(lo_, hi_);                                                        \
    }
        VDB_READ_B(fb1, img, 1)
        VDB_STEP(0, fa0, fb0, fa1, img, 1, true, ) VDB_STEP(1, fa0, fb0, fa1, img, 1, true, )
        VDB_STEP(2, fa0, fb0, fa1, img, 1, true, ) VDB_STEP(3, fa0, fb0, fa1, img, 1, true, )
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
        const uint32_t la_ = (4 * w) * 1024;
        const uint32_t lb_ = A_BYTES + (2 * w) * 1024;
        const uint32_t ka_ = fks * (KSTAGE * 4);
        const uint32_t kb_ = fks * B_BYTES;
#define VDB_PIECE_A(J_)                                                                                \
    if (do_dma && !(p.ablate & 2u)) { VDB_DMA_NT(a_piece(J_) + ka_, img, la_ + (J_) * 1024); }         \
    __builtin_amdgcn_sched_barrier(0);
#define VDB_PIECE_B(J_)                                                                                \
    if (do_dma && !(p.ablate & 4u)) { VDB_DMA(bbase + (ob[J_] + kb_), img, lb_ + (J_) * 1024); }       \
    __builtin_amdgcn_sched_barrier(0);
        if (more) VDB_READ_B(fb0, nxt, 0)
        VDB_STEP(0, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_A(0) VDB_PIECE_A(1))
        VDB_STEP(1, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_A(2) VDB_PIECE_A(3))
        VDB_STEP(2, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_B(0))
        VDB_STEP(3, fa1, fb1, fa0, nxt, 0, more, VDB_PIECE_B(1))
#undef VDB_PIECE_A
#undef VDB_PIECE_B
#undef VDB_STEP
#undef VDB_READ_B
        if (do_dma) VDB_ADV

        if (ks == KS - 1 && !(p.ablate & 8u)) {
            const uint32_t par = tile & 1u;
            // (the constants of this tile were issued at least one counted top-of-stage wait + barrier ago: every
            // stage that issues them either issues 6 row/query pieces after them or is followed by a vmcnt(0) wait)
            uint32_t tr0;                                               // device row of tile-row 0 (filter mode)
            uint32_t sj0 = 0;                                           // sample index of tile-row 0 (sample mode)
            if (SAMPLE) { sj0 = (tile_first + tile * tile_step) * TR; tr0 = 0; }
            else tr0 = r0 + tile * TS;
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
            // MARGIN: the filter is  lb = fma(-g_q, margin_row, score) <= thr.  Since margin_row <= mmax (the largest margin of
            // this wave's 128 rows), lb <= thr implies score <= thr + g_q mmax =: thp -- so the COMMON path compares the plain
            // score with a per-tile loosened threshold (two fmas per lane and tile instead of one packed fma and one more LDS
            // read per pair of elements), and only the rare path computes lb and applies the exact test.  The slack covers
            // the f32 rounding of thp and of lb, so no row with lb <= thr can fail the pre-test.
            const float* mg = sMarg + (MARGIN ? par * TR + wr * 128 : 0);
            float thp_a = thr_a, thp_b = thr_b, ng_a = 0.f, ng_b = 0.f;
            if (MARGIN) {
                float mm = fmaxf(mg[2 * lane], mg[2 * lane + 1]);       // +inf margins (norm overflow) open the tile; NaN rows carry NaN scores anyway
                for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
                const float ga = sG[q_a], gb = sG[q_b];
                ng_a = -ga; ng_b = -gb;
                thp_a = fmaf(ga, mm, thr_a); thp_a += (fabsf(thr_a) + ga * mm) * 6.0e-7f;
                thp_b = fmaf(gb, mm, thr_b); thp_b += (fabsf(thr_b) + gb * mm) * 6.0e-7f;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t vbits = (uint32_t)(val[i >> 1] >> (32 * (i & 1) + 4 * h));
                const uint32_t rowb = wr * 128 + i * 32 + 4 * h;       // tile-row of element (j = 0, e = 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 a4 = *reinterpret_cast<const float4*>(al + i * 32 + 8 * j);
                    const float4 b4 = *reinterpret_cast<const float4*>(be + i * 32 + 8 * j);
                    // scores of 4 rows x 2 queries
                    // (two rows per v_pk_fma_f32: the same IEEE fma per element, half the instructions)
                    const f32x2 al01 = {a4.x, a4.y}, al23 = {a4.z, a4.w}, be01 = {b4.x, b4.y}, be23 = {b4.z, b4.w};
                    const f32x2 pa01 = {acc[i][0][4 * j + 0], acc[i][0][4 * j + 1]}, pa23 = {acc[i][0][4 * j + 2], acc[i][0][4 * j + 3]};
                    const f32x2 pb01 = {acc[i][1][4 * j + 0], acc[i][1][4 * j + 1]}, pb23 = {acc[i][1][4 * j + 2], acc[i][1][4 * j + 3]};
                    const f32x2 ra01 = __builtin_elementwise_fma(pa01, al01, be01), ra23 = __builtin_elementwise_fma(pa23, al23, be23);
                    const f32x2 rb01 = __builtin_elementwise_fma(pb01, al01, be01), rb23 = __builtin_elementwise_fma(pb23, al23, be23);
                    const float sa0 = ra01.x, sa1 = ra01.y, sa2 = ra23.x, sa3 = ra23.y;
                    const float sb0 = rb01.x, sb1 = rb01.y, sb2 = rb23.x, sb3 = rb23.y;
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
                        // Hits are rare (about 0.1 % of the elements).  ONE compare per query for the four rows: the smallest of the four
                        // scores against the threshold, its lane mask straight into the not-taken branch; the append code is out of
                        // line.  v_min_f32 drops a NaN operand and a NaN score must pass (flat_index.rs:62) -- so this form is used
                        // as it stands only when no score of the launch can be NaN (fused_no_nan: every norm within
                        // [2^-40, 2^40]); otherwise a NaN-propagating sum of the four is tested as well (inf - inf gives a
                        // false alarm, which the exact per-row test of the rare path sorts out).
                        const f32x2 na_ = __builtin_elementwise_min(ra01, ra23), nb_ = __builtin_elementwise_min(rb01, rb23);
                        unsigned long long ma = __builtin_amdgcn_ballot_w64(!(fminf(na_.x, na_.y) > thp_a));
                        unsigned long long mb = __builtin_amdgcn_ballot_w64(!(fminf(nb_.x, nb_.y) > thp_b));
                        if (__builtin_expect(!no_nan, 0)) {                  // a real (wave-uniform) branch: the empty asm keeps hipcc from
                            asm volatile("" ::: "memory");                   // computing the sums always and selecting with v_cndmask
                            const f32x2 ua_ = ra01 + ra23, ub_ = rb01 + rb23;
                            const float ta_ = ua_.x + ua_.y, tb_ = ub_.x + ub_.y;
                            ma |= __builtin_amdgcn_ballot_w64(ta_ != ta_);
                            mb |= __builtin_amdgcn_ballot_w64(tb_ != tb_);
                        }
                        if (kDiag && (p.ablate & 4096u)) {                  // diag 4096: round 1's four compares per query (A/B)
                            ma = __builtin_amdgcn_ballot_w64(!(sa0 > thp_a) || !(sa1 > thp_a) || !(sa2 > thp_a) || !(sa3 > thp_a));
                            mb = __builtin_amdgcn_ballot_w64(!(sb0 > thp_b) || !(sb1 > thp_b) || !(sb2 > thp_b) || !(sb3 > thp_b));
                        }
                        // The append path is what the epilogue costs (with thresholds that let nothing pass the kernel is as
                        // fast as without an epilogue), so it is kept short: one 4-bit hit mask per lane and query, then a
                        // loop over its set bits -- typically one lane, one iteration -- instead of four masked regions.
#define VDB_APPEND(S0, S1, S2, S3, THP, THR, NG, POOL, PCNT)                                           \
    {                                                                                                  \
        uint32_t hm_ = (!((S0) > (THP)) ? 1u : 0u) | (!((S1) > (THP)) ? 2u : 0u) | (!((S2) > (THP)) ? 4u : 0u) | (!((S3) > (THP)) ? 8u : 0u); \
        hm_ &= (vbits >> (8 * j)) & 0xfu;                                                              \
        while (hm_) {                                                                                  \
            const uint32_t e_ = (uint32_t)__builtin_ctz(hm_);                                          \
            hm_ &= hm_ - 1u;                                                                           \
            float sc_ = e_ == 0 ? (S0) : e_ == 1 ? (S1) : e_ == 2 ? (S2) : (S3);                       \
            if (MARGIN) {                                              /* the exact test, on the lower-bound score */ \
                sc_ = fmaf((NG), mg[i * 32 + 8 * j + 4 * h + e_], sc_);                                \
                if (sc_ > (THR)) continue;                                                             \
            }                                                                                          \
            if (!(kDiag && (p.ablate & 32u)) && PCNT < p.capl) POOL[PCNT] = make_raw_key(sc_, tr0 + rt0 + e_); /* diag 32: count only */ \
            ++PCNT;                                                                                    \
        }                                                                                              \
    }
                        if (kDiag && (p.ablate & 64u)) {            // diag 64: the branch is taken, the append is not executed
                            if (__builtin_expect(ma != 0ull, 0)) { asm volatile("s_nop 1" ::: "memory"); ++pcnt_a; }
                            if (__builtin_expect(mb != 0ull, 0)) { asm volatile("s_nop 1" ::: "memory"); ++pcnt_b; }
                        } else {
                        if (__builtin_expect(ma != 0ull, 0)) VDB_APPEND(sa0, sa1, sa2, sa3, thp_a, thr_a, ng_a, pool_a, pcnt_a)
                        if (__builtin_expect(mb != 0ull, 0)) VDB_APPEND(sb0, sb1, sb2, sb3, thp_b, thr_b, ng_b, pool_b, pcnt_b)
                        }
#undef VDB_APPEND
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
    if (!SAMPLE && !(kDiag && (p.ablate & 2048u))) {                    // diag 2048: the counts are not written (cost of these stores)
        p.pool_cnt[sub_a] = pcnt_a;
        p.pool_cnt[sub_b] = pcnt_b;
    }
#undef VDB_DMA
#undef VDB_DMA_NT
#undef VDB_DMA4
#undef VDB_ISSUE
#undef VDB_ADV
#undef VDB_LOAD_FRAGS
}

void launch_fused_bf16p(const FusedBf16Params& p, hipStream_t s) {
    if (p.margin) hipLaunchKernelGGL((fused_bf16p_kernel<false, true>), dim3(p.n_wg), dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((fused_bf16p_kernel<false, false>), dim3(p.n_wg), dim3(NT), 0, s, p);
}

}  // namespace vdb
