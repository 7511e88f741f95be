// kernels_fused_a16.hip -- the screening kernel of kernels_fused_bf16.hip reading a bf16 SHADOW COPY of the rows
// (vdb_flat_set_shadow: +2 bytes per element of HBM, opt-in) instead of the f32 rows: half the bytes per pass.  The
// shadow is the RNE rounding of the f32 rows (v_cvt_pk_bf16_f32 at upload) -- exactly what the f32-row kernel computes in
// registers -- so scores, candidates, certificates and results are bit-identical to that kernel's; only the bytes differ.
//
// A stage is 32 K-elements of 256 rows and of 256 queries (bf16, 16 KB each).  Two things shape the fetch:
//  * The rows come from HBM, and the memory system retires REQUESTS, not bytes: with 64-byte row segments (one K stage
//    of a bf16 row) the pass ran at the same ~55 G requests/s as the f32-row kernel and therefore at half its bytes/s.
//    So the rows are fetched TWO K stages at a time -- 128 contiguous bytes per row, one full cache line per request --
//    into 32 KB row images that each serve two stages.
//  * The queries come from the L2, and a wave's vmcnt counter retires its vector-memory operations IN ORDER: a wave that
//    fetched both kinds could never have more row stages outstanding than query stages.  So the two kinds are fetched
//    by DIFFERENT waves, each counting only its own kind:
//      waves 0-3: the rows, a ring of THREE 32 KB images (two double stages = 64 KB per CU in flight); 8 pieces per
//                 wave at every odd stage, `s_waitcnt vmcnt(8)` before the barrier that publishes an even stage;
//      waves 4-7: the queries, a ring of THREE 16 KB images (two stages in flight, `vmcnt(4)`), and the per-row
//                 constants of the next tile.
//  * One barrier per stage, in the MIDDLE of the stage (see the stage loop): the LDS fragment reads run under the
//    MFMAs instead of in front of them, and the DMA refills the images of the stage being finished.
// All eight waves compute.  Fragments are read from LDS as bf16 (one ds_read_b128 each, no conversion in the loop).
// Tile shape, wave layout, epilogue and sample mode are the f32-row kernel's.  Requires an even number of K stages
// per row (padded dimension a multiple of 64); the host uses the f32-row kernel otherwise.
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
constexpr int A_ROWB = 128;                      // 64 bf16 per row: TWO K stages per row image
constexpr int B_ROWB = 64;                       // 32 bf16 per query and stage
constexpr int A_BYTES = TR * A_ROWB;             // 32 KB
constexpr int B_BYTES = TQ * B_ROWB;             // 16 KB
constexpr int MT = 4, QT = 2;                    // MFMA tiles per wave: 4 x 32 rows, 2 x 32 queries

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

}  // namespace

template <bool SAMPLE>
__global__ __launch_bounds__(NT, 2) void fused_a16_kernel(FusedBf16Params p) {
    // DISTINCT LDS objects, each access names its image at compile time
    __shared__ __attribute__((aligned(16))) char sA0[A_BYTES];
    __shared__ __attribute__((aligned(16))) char sA1[A_BYTES];
    __shared__ __attribute__((aligned(16))) char sA2[A_BYTES];
    __shared__ __attribute__((aligned(16))) char sB0[B_BYTES];
    __shared__ __attribute__((aligned(16))) char sB1[B_BYTES];
    __shared__ __attribute__((aligned(16))) char sB2[B_BYTES];
    // per-row constants of a tile (alpha, beta, the row's eligibility-mask word), double buffered by tile parity;
    // filled by LDS-DMA as well, so that no wave ever holds a pending ordinary load inside the stage loop
    __shared__ __attribute__((aligned(16))) float sAlpha[2 * TR];
    __shared__ __attribute__((aligned(16))) float sBeta[2 * TR];
    __shared__ __attribute__((aligned(16))) uint32_t sMaskW[2 * TR];
    // sample mode: the device rows of the (single) tile's scattered sample rows, kept here instead of in four 64-bit
    // address registers per lane (with those the sample instance spilled to scratch, and every scratch reload is a
    // vmcnt(0) -- a drain of the DMA pipeline -- in the stage loop); each lane reads back only what it wrote
    __shared__ uint32_t sRow[SAMPLE ? 32 * 64 : 1];

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
        sub_a = (((size_t)q_a * p.n_wg + blockIdx.x) * 2 + wr) * 2 + h;
        sub_b = (((size_t)q_b * p.n_wg + blockIdx.x) * 2 + wr) * 2 + h;
        // The pool KEYS are laid out workgroup-major -- slot ((wg*256 + q)*4 + row half*2 + lane half)*capl -- so that the
        // few scattered appends of one workgroup fall into ONE 2 MB region instead of one region per query (256 regions
        // 2 MB apart: every append then missed the CU's address-translation cache in front of the row stream).  The
        // counts stay query-major (pool_cnt[sub]); the select's gather knows both layouts (SelectParams::wg_major).
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
    const char* __restrict__ rows_b = reinterpret_cast<const char*>(p.rows16);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qb);

    // ---- DMA plan.  A row image = 32 pieces of 1 KB (8 rows x 128 B each: lane L -> row L>>3, 16-byte position L&7),
    // a query image = 16 pieces (16 queries x 64 B: lane L -> query L>>2, position L&3).  Row wave w (0-3) fills row
    // pieces 8w..8w+7, query wave w (4-7) query pieces 4(w-4)..+3.  Both images are XOR-swizzled so that the fragment
    // reads below are bank-conflict free: data chunk x of row r sits at position x ^ ((r>>1)&7), data chunk x of query r
    // at position x ^ ((r>>2)&3); the filling lane fetches the permuted source chunk.
    const bool row_wave = w < 4;                                        // wave-uniform
    const uint32_t a_pr = lane >> 3, a_pp = lane & 7;
    // tile-row of piece j: rt = 64w + 8j + a_pr, so (rt>>1)&7 = (4(j&1) + (a_pr>>1)) & 7: one source chunk for even j, one for odd j
    const uint32_t a_chunk0 = (a_pp ^ ((a_pr >> 1) & 7)) * 16, a_chunk1 = (a_pp ^ ((4 + (a_pr >> 1)) & 7)) * 16;
    // (the queries are stored by query_prep in exactly this image order, one 16 KB image per K stage: a wave's
    // query piece is 1 KB of CONTIGUOUS global memory -- 8 full 128-byte requests instead of 16 scattered 64-byte ones)
    const uint32_t ob = (4 * (w & 3)) * 1024 + lane * 16;
    // filter mode: the tile's rows are contiguous (the store is allocated and zero-filled in multiples of 256 rows, so
    // tile rows past the last row are readable; the eligibility ballots of the epilogue keep them out): pieces j, j+2
    // are 16 rows apart -> two base pointers (even j, odd j) and a uniform stride
    const char* aptr0 = nullptr; const char* aptr1 = nullptr;
    const size_t a_pair_stride = (size_t)16 * ld * 2;
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
        if (SAMPLE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sRow[(8 * (w & 3) + j) * 64 + lane] = tile_rows_of(t, 64 * (w & 3) + 8 * j + a_pr);
        } else {
            const uint32_t row = r0 + t * TR + 64 * (w & 3) + a_pr;    // unclamped, see above
            aptr0 = rows_b + (size_t)row * ld * 2 + a_chunk0;
            aptr1 = rows_b + (size_t)(row + 8) * ld * 2 + a_chunk1;
        }
    };
    auto a_piece = [&](int j) -> const char* {
        if (SAMPLE) return rows_b + (size_t)sRow[(8 * (w & 3) + j) * 64 + lane] * ld * 2 + ((j & 1) ? a_chunk1 : a_chunk0);
        return ((j & 1) ? aptr1 : aptr0) + (size_t)(j >> 1) * a_pair_stride;
    };
    // The LDS-DMA is issued from inline asm, not through __builtin_amdgcn_global_load_lds (see kernels_fused_bf16.hip):
    // all ordering between the DMA and the LDS reads is done by hand (counted s_waitcnt + s_barrier at the top of
    // each stage); compiler-inserted vmcnt waits for ordinary loads stay correct because not counting these
    // instructions only makes them wait longer.
#define VDB_DMA(GP, IMG, LOFF)                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                     \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
    // rows are read once per launch: non-temporal, so that they do not push the queries out of the L2
#define VDB_DMA_NT(GP, IMG, LOFF)                                                                      \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt"                  \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)((IMG) + (LOFF))), "v"((const void*)(GP)) : "memory", "m0")
#define VDB_ISSUE_A(IMG, SSI)                                                                          \
    {                                                                                                  \
        const uint32_t la_ = (8 * (w & 3)) * 1024;                                                     \
        const uint32_t ka_ = (SSI) * A_ROWB;                                                           \
        if (!(p.ablate & 2u)) {                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                             \
            const char* g_ = a_piece(j_) + ka_;                                                        \
            VDB_DMA_NT(g_, IMG, la_ + j_ * 1024);                                                      \
        }                                                                                              \
        }                                                                                              \
    }
#define VDB_ISSUE_B(IMG, KSI)                                                                          \
    {                                                                                                  \
        const uint32_t lb_ = (4 * (w & 3)) * 1024;                                                     \
        const uint32_t kb_ = (KSI) * B_BYTES + ob;                                                     \
        if (!(p.ablate & 4u)) {                                                                        \
        VDB_DMA(bbase + kb_, IMG, lb_);                                                                \
        VDB_DMA(bbase + (kb_ + 1024), IMG, lb_ + 1024);                                                \
        VDB_DMA(bbase + (kb_ + 2048), IMG, lb_ + 2048);                                                \
        VDB_DMA(bbase + (kb_ + 3072), IMG, lb_ + 3072);                                                \
        }                                                                                              \
    }

    // ---- row constants of a tile, one tile ahead, by LDS-DMA (4 bytes per lane): query wave w fetches alpha, beta and
    // the mask word of rows 64(w-4)..+63.  Issued BEFORE that stage's query pieces, so the counted wait of the query
    // waves at the top of the next stage covers them, whatever the number of K stages per tile.
#define VDB_DMA4(GP, LP)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"                        \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)(LP)), "v"((const void*)(GP)) : "memory", "m0")
    auto issue_consts = [&](uint32_t t) {                               // query waves only
        const uint32_t par = t & 1u;
        const uint32_t cr = 64 * (w & 3);                              // first tile-row of this wave's chunk
        const uint32_t row = tile_rows_of(t, cr + lane);
        VDB_DMA4(p.alpha + row, sAlpha + par * TR + cr);
        VDB_DMA4(p.beta + row, sBeta + par * TR + cr);
        VDB_DMA4(p.rowmask + (row >> 5), sMaskW + par * TR + cr);
    };

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
    const uint32_t b_row_off = (wq * 64 + c) * B_ROWB;                  // + j*32*B_ROWB
    uint32_t ra[2][2], rb[2];                                           // [stage half of the row image][k-step], [k-step]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ra[0][t] = ((2 * t + h) ^ swa) * 16;
        ra[1][t] = ((4 + 2 * t + h) ^ swa) * 16;
        rb[t] = ((2 * t + h) ^ swb) * 16;
    }

    // ---- prologue: row waves put double stages 0 and 1 in flight, query waves the constants of tile 0 and stages 0, 1
    uint32_t tile = 0, ks = 0;                                          // of the stage being computed
    uint32_t ftile = 0, fss = 0;                                        // of the next row double stage to fetch
    uint32_t bks = 0;                                                   // K stage of the next query stage to fetch
    const uint32_t SS = KS >> 1;                                        // double stages per tile (KS is even)
#define VDB_ADV_A                                                                                      \
    {                                                                                                  \
        ++fss;                                                                                         \
        if (fss == SS) { fss = 0; ++ftile; if (ftile < ntiles) set_tile_ptrs(ftile); }                 \
    }
#define VDB_ADV_B { ++bks; if (bks == KS) bks = 0; }
    if (row_wave) {
        set_tile_ptrs(0);
        VDB_ISSUE_A(sA0, fss) VDB_ADV_A
        if (total > 2) { VDB_ISSUE_A(sA1, fss) VDB_ADV_A }
        if (total > 4) { VDB_ISSUE_A(sA2, fss) VDB_ADV_A }
        if (total > 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (total > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_consts(0);
        VDB_ISSUE_B(sB0, bks) VDB_ADV_B
        VDB_ISSUE_B(sB1, bks) VDB_ADV_B                                 // total >= 2
        if (total > 2) { VDB_ISSUE_B(sB2, bks) VDB_ADV_B }
        if (total > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                       // stage 0 is published
    asm volatile("" ::: "memory");

    // The stage loop is software-pipelined around ONE barrier per stage that sits in the MIDDLE of the stage:
    //     read the fragments of k-step 1 of stage s            (LDS, overlaps the next line)
    //     8 MFMAs of k-step 0                                   (fragments read during the previous stage)
    //     wait for this wave's pieces of stage s+1, barrier     -> stage s+1 is published, and nobody reads the images of
    //                                                              stage s any more (every wave waited for its LDS reads)
    //     issue the DMA of the next stages INTO THE IMAGES OF STAGE s
    //     read the fragments of k-step 0 of stage s+1           (LDS, overlaps the next line)
    //     8 MFMAs of k-step 1
    // so the LDS fragment traffic (96 KB per stage and CU, ~770 clocks of LDS bandwidth) runs under the ~1000 clocks of
    // MFMA work instead of in front of it, and the ring is one stage deeper for the same LDS: three row images hold the
    // double stage being computed and two in flight, three query images the stage being computed and two in flight.
    bf16x8 fa0[MT], fb0[QT];                                            // k-step 0 fragments of the stage to compute next
#define VDB_LOAD_FRAGS(FA, FB, IA, IB, U_, T_)                                                         \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) {                                            \
            const u32x4 raw_ = *reinterpret_cast<const u32x4*>((IA) + a_row_off + i_ * 32 * A_ROWB + ra[U_][T_]); \
            FA[i_] = __builtin_bit_cast(bf16x8, raw_);                                                 \
        }                                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < QT; ++j_) {                                            \
            const u32x4 raw_ = *reinterpret_cast<const u32x4*>((IB) + b_row_off + j_ * 32 * B_ROWB + rb[T_]); \
            FB[j_] = __builtin_bit_cast(bf16x8, raw_);                                                 \
        }                                                                                              \
    }
#define VDB_MFMAS(FA, FB)                                                                              \
    {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                              \
            _Pragma("unroll") for (int j_ = 0; j_ < QT; ++j_)                                          \
                acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i_], FB[j_], acc[i_][j_], 0, 0, 0); \
    }
    VDB_LOAD_FRAGS(fa0, fb0, sA0, sB0, 0, 0)

    // STEADY: the caller guarantees st + 5 < total, so the waits and the DMA issue are unconditional (see the f32-row
    // kernel for why that matters to hipcc's waitcnt pass).
    auto run_stage = [&](uint32_t st, auto pos_tag, auto steady_tag) {
        constexpr int K6 = decltype(pos_tag)::value;                    // st mod 6
        constexpr int AB = K6 >> 1;                                     // row image (two stages each)
        constexpr int U = K6 & 1;                                       // which half of the row image
        constexpr int BB = K6 % 3;                                      // query image
        constexpr int N6 = (K6 + 1) % 6;                                // the same for stage st + 1
        constexpr int NAB = N6 >> 1, NU = N6 & 1, NBB = N6 % 3;
        constexpr bool STEADY = decltype(steady_tag)::value;
        char* imgA = AB == 0 ? sA0 : AB == 1 ? sA1 : sA2;
        char* imgB = BB == 0 ? sB0 : BB == 1 ? sB1 : sB2;
        const char* nxtA = NAB == 0 ? sA0 : NAB == 1 ? sA1 : sA2;
        const char* nxtB = NBB == 0 ? sB0 : NBB == 1 ? sB1 : sB2;
        bf16x8 fa1[MT], fb1[QT];
        VDB_LOAD_FRAGS(fa1, fb1, imgA, imgB, U, 1)
        VDB_MFMAS(fa0, fb0)
        // publish stage st+1: a row wave's pieces of the next row image have landed once at most the 8 pieces of the
        // double stage after it are outstanding (an even stage's successor reads the same image), a query wave's
        // once at most the 4 pieces of stage st+2 are.  lgkmcnt(0): this wave's fragment reads of stage st are done,
        // so after the barrier the images of stage st are free.  (A bare s_barrier: __syncthreads() carries a fence that
        // hipcc lowers to vmcnt(0), which would drain the DMA pipeline at every stage; the asm memory clobbers keep the
        // compiler from moving LDS accesses across.)
        if (row_wave) {
            if (U == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (STEADY || st + 3 < total) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            if (STEADY || st + 2 < total) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // The DMA instructions are issued BETWEEN the MFMAs of k-step 1, one piece behind each MFMA: right after the
        // barrier all eight waves are in the same phase, and eight waves issuing 4-8 vector-memory instructions each
        // before their first MFMA left the matrix pipe idle for that long in every stage.
        // ONE straight-line MFMA sequence (any second copy of it on another control-flow path makes hipcc shuffle the 128
        // accumulator registers between paths and spill); only the DMA instructions sit under (wave-uniform) branches.
        // Every wave issues four pieces per stage: a row wave the first half of double stage S+3 at the odd stage that
        // frees the image, the second half at the even stage after it; a query wave the four pieces of stage st+3.
        if (STEADY || st + 1 < total) VDB_LOAD_FRAGS(fa0, fb0, nxtA, nxtB, NU, 0)
        char* prvA = AB == 0 ? sA2 : AB == 1 ? sA0 : sA1;               // the row image of the previous double stage
        const bool do_a = row_wave && !(p.ablate & 2u) && (U == 1 ? (STEADY || st + 5 < total) : (st > 0 && (STEADY || st + 4 < total)));
        const bool do_b = !row_wave && !(p.ablate & 4u) && (STEADY || st + 3 < total);
        const uint32_t la_ = (8 * (w & 3) + (U == 1 ? 0 : 4)) * 1024;
        const uint32_t ka_ = fss * A_ROWB;
        const uint32_t lb_ = (4 * (w & 3)) * 1024;
        const uint32_t kb_ = bks * B_BYTES + ob;
        // constants of the NEXT tile into the other parity (every wave is past the epilogue that read it)
        if (!row_wave && ks == 0 && tile + 1 < ntiles) issue_consts(tile + 1);
#define VDB_MFMA1(M_)                                                                                  \
    acc[(M_) >> 1][(M_) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[(M_) >> 1], fb1[(M_) & 1], acc[(M_) >> 1][(M_) & 1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
#define VDB_PIECE(J_)                                                                                  \
    if (do_a) {                                                                                        \
        const char* g_ = a_piece((U == 1 ? 0 : 4) + (J_)) + ka_;                                       \
        VDB_DMA_NT(g_, (U == 1 ? imgA : prvA), la_ + (J_) * 1024);                                     \
    } else if (do_b) {                                                                                 \
        VDB_DMA(bbase + (kb_ + (J_) * 1024), imgB, lb_ + (J_) * 1024);                                 \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
        VDB_MFMA1(0) VDB_PIECE(0) VDB_MFMA1(1) VDB_MFMA1(2) VDB_PIECE(1) VDB_MFMA1(3)
        VDB_MFMA1(4) VDB_PIECE(2) VDB_MFMA1(5) VDB_MFMA1(6) VDB_PIECE(3) VDB_MFMA1(7)
#undef VDB_PIECE
#undef VDB_MFMA1
        if (row_wave) {
            if (U == 0 && st > 0 && (STEADY || st + 4 < total)) VDB_ADV_A      // both halves of the double stage are issued
        } else {
            if (STEADY || st + 3 < total) VDB_ADV_B
        }

        if (ks == KS - 1 && !(p.ablate & 8u)) {
            const uint32_t par = tile & 1u;
            // (the constants of this tile were issued by the query waves before the query pieces of a stage whose
            // top-of-stage wait + barrier every wave has passed, so they have landed)
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
                        // Hits are rare (about 0.2 % of the elements).  Common path per query: four compares whose
                        // lane masks are OR-ed on the scalar unit and ONE not-taken branch; the append code is out of
                        // line.  `!(s > thr)` keeps a NaN score (it must reach the re-rank, flat_index.rs:62).
                        const unsigned long long ma = __builtin_amdgcn_ballot_w64(!(sa0 > thr_a)) | __builtin_amdgcn_ballot_w64(!(sa1 > thr_a)) |
                                                      __builtin_amdgcn_ballot_w64(!(sa2 > thr_a)) | __builtin_amdgcn_ballot_w64(!(sa3 > thr_a));
                        const unsigned long long mb = __builtin_amdgcn_ballot_w64(!(sb0 > thr_b)) | __builtin_amdgcn_ballot_w64(!(sb1 > thr_b)) |
                                                      __builtin_amdgcn_ballot_w64(!(sb2 > thr_b)) | __builtin_amdgcn_ballot_w64(!(sb3 > thr_b));
                        // The append path is what the epilogue costs (with thresholds that let nothing pass the kernel is as
                        // fast as without an epilogue), so it is kept short: one 4-bit hit mask per lane and query, then a
                        // loop over its set bits -- typically one lane, one iteration -- instead of four masked regions.
#define VDB_APPEND(S0, S1, S2, S3, THR, POOL, PCNT)                                                    \
    {                                                                                                  \
        uint32_t hm_ = (!((S0) > (THR)) ? 1u : 0u) | (!((S1) > (THR)) ? 2u : 0u) | (!((S2) > (THR)) ? 4u : 0u) | (!((S3) > (THR)) ? 8u : 0u); \
        hm_ &= (vbits >> (8 * j)) & 0xfu;                                                              \
        while (hm_) {                                                                                  \
            const uint32_t e_ = (uint32_t)__builtin_ctz(hm_);                                          \
            hm_ &= hm_ - 1u;                                                                           \
            const float sc_ = e_ == 0 ? (S0) : e_ == 1 ? (S1) : e_ == 2 ? (S2) : (S3);                 \
            if (PCNT < p.capl) POOL[PCNT] = make_raw_key(sc_, tr0 + rt0 + e_);                         \
            ++PCNT;                                                                                    \
        }                                                                                              \
    }
                        if (__builtin_expect(ma != 0ull, 0)) VDB_APPEND(sa0, sa1, sa2, sa3, thr_a, pool_a, pcnt_a)
                        if (__builtin_expect(mb != 0ull, 0)) VDB_APPEND(sb0, sb1, sb2, sb3, thr_b, pool_b, pcnt_b)
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

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    uint32_t st = 0;
    for (; st + 10 < total; st += 6) {                                   // (stage index mod 6) >> 1 and mod 3 == image indices
        run_stage(st, I0{}, std::true_type{});
        run_stage(st + 1, I1{}, std::true_type{});
        run_stage(st + 2, I2{}, std::true_type{});
        run_stage(st + 3, I3{}, std::true_type{});
        run_stage(st + 4, I4{}, std::true_type{});
        run_stage(st + 5, I5{}, std::true_type{});
    }
    // the last two to ten stages (total is even): conditional issue
    if (st < total) { run_stage(st, I0{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I1{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I2{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I3{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I4{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I5{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I0{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I1{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I2{}, std::false_type{}); ++st; }
    if (st < total) { run_stage(st, I3{}, std::false_type{}); ++st; }
    if (!SAMPLE) {
        p.pool_cnt[sub_a] = pcnt_a;
        p.pool_cnt[sub_b] = pcnt_b;
    }
#undef VDB_DMA
#undef VDB_DMA_NT
#undef VDB_DMA4
#undef VDB_ISSUE_A
#undef VDB_ISSUE_B
#undef VDB_ADV_A
#undef VDB_ADV_B
#undef VDB_LOAD_FRAGS
#undef VDB_MFMAS
}

// same tile shape, sub-pool and sample-group layout as the f32-row kernel (fused_bf16_tile_rows & co.)
void launch_fused_a16(const FusedBf16Params& p, hipStream_t s) {
    hipLaunchKernelGGL(fused_a16_kernel<false>, dim3(p.n_wg), dim3(NT), 0, s, p);
}
// (the sample pass -- 65536 rows, a tenth of a millisecond's worth of the f32-row kernel -- keeps reading the f32 rows:
// the <true> instance of this kernel needs more than the 256 registers a wave has here and would spill into the stage
// loop; its thresholds are the same either way, so the candidates are identical too)

}  // namespace vdb
