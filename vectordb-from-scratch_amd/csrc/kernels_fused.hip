// kernels_fused.hip -- the hot kernel: batched query x database inner products on
// v_mfma_f32_32x32x2_f32 (exact f32 in / f32 accumulate), fused with the ranking-score
// epilogue and an inclusive threshold filter that appends the few surviving
// (score, row) keys to a per-query candidate pool.  The B x N score matrix never
// exists in memory.   gfx950 (MI355X, CDNA4) only.
//
// Workgroup = 256 threads = 4 waves (one per SIMD); TWO workgroups are resident per CU
//   (75 KB LDS, <= 256 VGPRs each) and run out of phase: while one is in its staging /
//   barrier / epilogue phase the other keeps the SIMD's MFMA pipe busy.  (A single
//   8-wave workgroup per CU measured 73 % MFMA-pipe utilisation: its two waves per
//   SIMD hit every barrier together.  profiles/r01_a_pmc_fused_v1.json)
// Tile      = 128 database rows x (32*NQT) queries, K staged 32 floats at a time.
//   wave w: query tile qt = w % NQT, row part rp = w / NQT, MT accumulator tiles of
//           32x32 (rows (rp*MT+i)*32.., i < MT).  Headline shape NQT = 4, MT = 4:
//           128 rows x 128 queries, 64 accumulator VGPRs per lane.
//   MFMA operand map (32x32x2): A[i=lane&31][k=lane>>5] = database row, B[k][j=lane&31]
//   = query, so every lane owns ONE query column (its threshold lives in a register) and
//   16 rows per tile: row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// K order inside a stage: 4 groups of 8; lanes 0-31 feed k = 8g+s, lanes 32-63 feed
//   k = 8g+4+s (one 16-byte LDS read gives a lane its operand for 4 MFMAs).  The dense
//   sample kernel uses the same order, so scores are bit-identical between the two.
// LDS rows are padded 128 -> 144 bytes: ds_read_b128 of 16 distinct rows (mod 16) then
//   hits 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// Pipeline: register-staged double buffer, one barrier per K stage; global loads of
//   stage s+1 are issued before the MFMAs of stage s and written to LDS after them.
// Work split: grid.x persistent row ranges (multiples of 32 rows, equal within 32),
//   grid.y query super-tiles; a partial last tile only issues MFMAs for its valid
//   32-row blocks.
#include "kernels.h"

#include <type_traits>

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROWB = 144;                 // padded LDS bytes per 32-float row

// NW  = waves per workgroup (4: two independent workgroups share a CU, one wave each per SIMD)
// NQT = 32-query tiles per workgroup; MT = 32-row accumulator tiles per wave.
// RP = NW/NQT waves share a query tile and split the tile's rows; TR = 32*MT*RP rows per tile.
template <int NQT, int MT, int NW> struct FusedCfg {
    static constexpr int NT = NW * 64;
    static constexpr int RP = NW / NQT;
    static constexpr int TR = 32 * MT * RP;
    static constexpr int A_BYTES = TR * ROWB;
    static constexpr int B_BYTES = 32 * NQT * ROWB;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int CONST_OFF = 2 * STAGE_BYTES;           // alpha[3][TR] beta[3][TR] valid[3][TR/32]
    static constexpr int LDS_BYTES = CONST_OFF + 3 * TR * 4 * 2 + 3 * (TR / 32) * 4;
    static constexpr int NA = (TR * 8) / NT;                    // A float4 loads per thread per stage
    static constexpr int NB = (NQT * 256 + NT - 1) / NT;        // B float4 loads per thread per stage
    static_assert(NA >= 1 && NA <= 4 && NB >= 1 && NB <= 4 && (NQT * 256) % NT == 0 && TR % 64 == 0 && TR <= NT, "unsupported shape");
};
// shipped shapes (all 128-row tiles, 4 waves): queries per workgroup 128 / 64 / 32
using CfgQ128 = FusedCfg<4, 4, 4>;   // headline: 128 queries x 128 rows, 64 accumulator VGPRs per lane, 75 KB LDS
using CfgQ64 = FusedCfg<2, 2, 4>;
using CfgQ32 = FusedCfg<1, 1, 4>;
using CfgQ256 = FusedCfg<8, 4, 8>;   // alternative: ONE 8-wave workgroup per CU, 256 queries share each fetched row tile

size_t fused_lds_bytes(int nqt) {
    switch (nqt) {
    case 1: return CfgQ32::LDS_BYTES;
    case 2: return CfgQ64::LDS_BYTES;
    case 8: return CfgQ256::LDS_BYTES;
    default: return CfgQ128::LDS_BYTES;
    }
}
uint32_t fused_tile_rows(int) { return 128; }

#define VDB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define VDB_PIN() do { if (VDB_USE_PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef VDB_USE_PIN
#define VDB_USE_PIN 0
#endif

template <int NQT, int MT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void fused_score_filter_kernel(FusedParams p) {
    using C = FusedCfg<NQT, MT, NW>;
    constexpr int TR = C::TR;
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per-tile row constants, 3 rotating slots: the slot of tile t+1 is written during tile t's last
    // stage while tile t's slot is still to be read by its epilogue; three slots keep that safe even
    // when a tile is a single K stage
    float* sAlpha = reinterpret_cast<float*>(smem + C::CONST_OFF);      // [3][TR]
    float* sBeta = sAlpha + 3 * TR;                                     // [3][TR]
    uint32_t* sValid = reinterpret_cast<uint32_t*>(sBeta + 3 * TR);     // [3][TR/32]

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t qt = w % NQT, rp = w / NQT;

    // ---- this workgroup's row range (multiples of 32 rows) and query super-tile
    const uint32_t nblk = (p.n_rows + 31) >> 5;
    const uint32_t b0 = (uint32_t)(((uint64_t)blockIdx.x * nblk) / p.n_wg);
    const uint32_t b1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * nblk) / p.n_wg);
    const uint32_t r0 = b0 * 32;
    const uint32_t r1 = (b1 * 32 < p.n_rows) ? b1 * 32 : p.n_rows;
    const uint32_t qwg = p.q_base + blockIdx.y * (32 * NQT);            // first query of this workgroup
    const uint32_t q = qwg + qt * 32 + c;                              // this lane's query
    // private candidate sub-pool of this lane: [query][row range][row part][lane half][capl]
    const size_t sub = (((size_t)q * p.n_wg + blockIdx.x) * C::RP + rp) * 2 + h;
    uint64_t* mypool = p.pool + sub * p.capl;
    uint32_t pcnt = 0;                                                  // keys appended by this lane so far
    if (r0 >= r1) {                                                     // empty range: publish the zero count
        p.pool_cnt[sub] = 0;
        return;
    }
    const uint32_t ntiles = (r1 - r0 + TR - 1) / TR;
    const uint32_t KS = p.ld / KSTAGE;
    const uint32_t total = ntiles * KS;
    const float thrq = p.thr[q];
    const uint32_t ld = p.ld;
    const uint32_t last_row = p.n_rows - 1;

    // ---- staging registers (global -> VGPR -> LDS); named scalars, not arrays, so they stay in VGPRs.
    // ONE register set: during stage s, register i is written to LDS (stage s+1's data, loaded one
    // stage ago) and immediately re-loaded with stage s+2's data, in the shadow of the MFMAs.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float r_alpha = 0.f, r_beta = 0.f;
    uint32_t r_mask = 0, r_bit = 0;
    bool r_inrange = false;
    // this thread always stages chunk (tid&7) of rows (tid>>3) + i*NT/8
    const uint32_t srow = tid >> 3, schunk = (tid & 7) * 4;
    const uint32_t lds_st = srow * ROWB + (tid & 7) * 16;               // staging write offset inside a tile image
    // Addresses = wave-uniform base (SGPR pair) + 32-bit per-thread byte offset, so a staging load costs
    // one v_add.  The per-tile parts (oa*, oc_*) are recomputed only when the fetched tile changes.
    const char* __restrict__ abase = reinterpret_cast<const char*>(p.rows + (size_t)r0 * ld);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qp + (size_t)qwg * ld);
    const uint32_t rows_wg = last_row - r0;                             // clamp: rows past the index stay in bounds
    const uint32_t ob = (srow * ld + schunk) * 4;                       // B: fixed rows of the workgroup
    const uint32_t ob_step = (NT / 8) * ld * 4;
    uint32_t oa0 = 0, oa1 = 0, oa2 = 0, oa3 = 0;                        // A: byte offset of this thread's rows in tile `otile`
    uint32_t oc_row = 0;                                                // constants: row (relative to r0) of this thread
#define VDB_TILE_OFFSETS(TILE)                                                                         \
    {                                                                                                  \
        const uint32_t t0_ = (TILE) * TR + srow;                                                       \
        uint32_t q0_ = t0_, q1_ = t0_ + (NT / 8), q2_ = t0_ + 2 * (NT / 8), q3_ = t0_ + 3 * (NT / 8);  \
        q0_ = q0_ > rows_wg ? rows_wg : q0_; q1_ = q1_ > rows_wg ? rows_wg : q1_;                      \
        q2_ = q2_ > rows_wg ? rows_wg : q2_; q3_ = q3_ > rows_wg ? rows_wg : q3_;                      \
        oa0 = (q0_ * ld + schunk) * 4; oa1 = (q1_ * ld + schunk) * 4;                                  \
        oa2 = (q2_ * ld + schunk) * 4; oa3 = (q3_ * ld + schunk) * 4;                                  \
        const uint32_t cr_ = (TILE) * TR + (tid % TR);                                                 \
        r_inrange_next = r0 + cr_ < r1;                                                                \
        oc_row = r_inrange_next ? cr_ : rows_wg;                                                       \
    }
    bool r_inrange_next = false;
    // load register REG with its float4 of k-stage KSI of the tile whose offsets are current
#define VDB_LA(I, REG, OA, KSI)                                                                        \
    if constexpr (C::NA > (I)) { REG = *reinterpret_cast<const float4*>(abase + ((OA) + (KSI) * (KSTAGE * 4))); }
#define VDB_LB(I, REG, KSI)                                                                            \
    if constexpr (C::NB > (I)) { REG = *reinterpret_cast<const float4*>(bbase + (ob + (I) * ob_step + (KSI) * (KSTAGE * 4))); }
    // row constants of the tile whose offsets are current: loaded by every thread, every stage (thread t
    // and t+TR load the same row; all VMEM in the loop is unconditional so that hipcc counts vmcnt exactly)
#define VDB_LC()                                                                                       \
    {                                                                                                  \
        const uint32_t rr_ = r0 + oc_row;                                                              \
        r_inrange = r_inrange_next;                                                                    \
        r_bit = rr_ & 31;                                                                              \
        r_mask = p.rowmask[rr_ >> 5]; /* raw loads only: they are consumed one stage later (VDB_SC) */ \
        r_alpha = p.alpha[rr_];                                                                        \
        r_beta = p.beta[rr_];                                                                          \
    }
    // store register REG into the LDS image BUF
#define VDB_SA(I, REG, BUF)                                                                            \
    if constexpr (C::NA > (I)) {                                                                       \
        *reinterpret_cast<float4*>(smem + (BUF) * C::STAGE_BYTES + lds_st + (I) * (NT / 8) * ROWB) = REG; \
    }
#define VDB_SB(I, REG, BUF)                                                                            \
    if constexpr (C::NB > (I)) {                                                                       \
        *reinterpret_cast<float4*>(smem + (BUF) * C::STAGE_BYTES + C::A_BYTES + lds_st + (I) * (NT / 8) * ROWB) = REG; \
    }
#define VDB_SC(TILE, KSI)                                                                              \
    if ((KSI) == 0) { /* wave-uniform; LDS stores only */                                              \
        const uint32_t par_ = (TILE) % 3;                                                              \
        const bool ok_ = r_inrange && ((r_mask >> r_bit) & 1u);                                        \
        sAlpha[par_ * TR + (tid % TR)] = r_alpha;                                                      \
        sBeta[par_ * TR + (tid % TR)] = ok_ ? r_beta : __uint_as_float(0x7f800000u); /* +inf never passes */ \
        unsigned long long bal_ = __ballot(ok_);                                                       \
        if (w < TR / 64 && lane == 0) {                                                                \
            sValid[par_ * (TR / 32) + 2 * w] = (uint32_t)bal_;                                         \
            sValid[par_ * (TR / 32) + 2 * w + 1] = (uint32_t)(bal_ >> 32);                             \
        }                                                                                              \
    }

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // ---- prologue: stage 0 -> LDS image 0, stage 1 -> registers (in flight)
    VDB_TILE_OFFSETS(0u)
    VDB_LA(0, ra0, oa0, 0u) VDB_LA(1, ra1, oa1, 0u) VDB_LA(2, ra2, oa2, 0u) VDB_LA(3, ra3, oa3, 0u)
    VDB_LB(0, rb0, 0u) VDB_LB(1, rb1, 0u) VDB_LB(2, rb2, 0u) VDB_LB(3, rb3, 0u)
    VDB_LC()
    VDB_SA(0, ra0, 0u) VDB_SA(1, ra1, 0u) VDB_SA(2, ra2, 0u) VDB_SA(3, ra3, 0u)
    VDB_SB(0, rb0, 0u) VDB_SB(1, rb1, 0u) VDB_SB(2, rb2, 0u) VDB_SB(3, rb3, 0u)
    VDB_SC(0u, 0u)
    uint32_t tile = 0, ks = 0;             // stage being computed
    uint32_t tile1 = 0, ks1 = 1;           // stage st+1 (in the staging registers)
    if (ks1 == KS) { ks1 = 0; tile1 = 1; VDB_TILE_OFFSETS(1u) }
    VDB_LA(0, ra0, oa0, ks1) VDB_LA(1, ra1, oa1, ks1) VDB_LA(2, ra2, oa2, ks1) VDB_LA(3, ra3, oa3, ks1)
    VDB_LB(0, rb0, ks1) VDB_LB(1, rb1, ks1) VDB_LB(2, rb2, ks1) VDB_LB(3, rb3, ks1)
    VDB_LC()
    __syncthreads();

    // One K stage.  PARTIAL selects the code for a tile with fewer than TR valid rows (only the last
    // tile of a row range); it runs in a separate loop so that the hot loop has one straight-line
    // MFMA block (a shared block with a branch made the compiler copy all accumulators every stage).
    auto run_stage = [&](uint32_t st, auto partial_tag) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        const uint32_t tr0 = r0 + tile * TR;
        const uint32_t mt_valid = PARTIAL ? ((r1 - tr0 + 31) >> 5) : (uint32_t)(TR / 32);
        (void)st;
        uint32_t tile2 = tile1, ks2 = ks1 + 1;
        if (ks2 == KS) { ks2 = 0; ++tile2; VDB_TILE_OFFSETS(tile2) }   // wave-uniform, once per tile, VALU only
        const uint32_t nbuf = (st + 1) & 1;
        // staging step I: write one register of stage st+1 to the other LDS image, then refill it
        // (past the end of the range the stores go to an LDS image nobody reads and the loads are
        //  clamped to valid rows: keeping them unconditional is what lets the waits be counted)
#ifdef VDB_DIAG   /* diagnostic build: ablate bit 32 skips the staging stores, bit 64 the staging loads */
#define VDB_STEP_A(I, REG, OA) { if (!(p.ablate & 32u)) { VDB_SA(I, REG, nbuf) } if (!(p.ablate & 64u)) { VDB_LA(I, REG, OA, ks2) } }
#define VDB_STEP_B(I, REG) { if (!(p.ablate & 32u)) { VDB_SB(I, REG, nbuf) } if (!(p.ablate & 64u)) { VDB_LB(I, REG, ks2) } }
#define VDB_STEP_C() { if (!(p.ablate & 32u)) { VDB_SC(tile1, ks1) } if (!(p.ablate & 64u)) { VDB_LC() } }
#else
#define VDB_STEP_A(I, REG, OA) { VDB_SA(I, REG, nbuf) VDB_LA(I, REG, OA, ks2) }
#define VDB_STEP_B(I, REG) { VDB_SB(I, REG, nbuf) VDB_LB(I, REG, ks2) }
#define VDB_STEP_C() { VDB_SC(tile1, ks1) VDB_LC() }
#endif

        const char* sa = smem + (st & 1) * C::STAGE_BYTES;
        const char* bptr = sa + C::A_BYTES + (qt * 32 + c) * ROWB + h * 16;
        const char* aptr = sa + (rp * MT * 32 + c) * ROWB + h * 16;
        if (p.ablate & 1u) {
            VDB_STEP_C()
            VDB_STEP_A(0, ra0, oa0) VDB_STEP_A(1, ra1, oa1) VDB_STEP_A(2, ra2, oa2) VDB_STEP_A(3, ra3, oa3)
            VDB_STEP_B(0, rb0) VDB_STEP_B(1, rb1) VDB_STEP_B(2, rb2) VDB_STEP_B(3, rb3)
        } else if constexpr (!PARTIAL) {
            // fragments double-buffered across the 4 K groups: group g+1 is read while g's MFMAs run
            float4 fbA, fbB, faA[MT], faB[MT];
            fbA = *reinterpret_cast<const float4*>(bptr);
#pragma unroll
            for (int i = 0; i < MT; ++i) faA[i] = *reinterpret_cast<const float4*>(aptr + i * 32 * ROWB);
#define VDB_GROUP(G, FB, FA, FBN, FAN, S0, S1)                                                         \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(FA[i].x, FB.x, acc[i]);       \
        S0;                                                                                            \
        VDB_PIN();                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(FA[i].y, FB.y, acc[i]);       \
        S1;                                                                                            \
        if constexpr ((G) < 3) {                                                                       \
            FBN = *reinterpret_cast<const float4*>(bptr + ((G) + 1) * 32);                             \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                             \
                FAN[i] = *reinterpret_cast<const float4*>(aptr + i * 32 * ROWB + ((G) + 1) * 32);      \
        }                                                                                              \
        VDB_PIN();                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(FA[i].z, FB.z, acc[i]);       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(FA[i].w, FB.w, acc[i]);       \
        VDB_PIN();                                                                                     \
    }
            // the row-constant step goes FIRST: its loads are loop-carried scalars that hipcc copies at the
            // loop latch, so they need the rest of the stage to land before that copy's wait
            VDB_GROUP(0, fbA, faA, fbB, faB, VDB_STEP_C() VDB_STEP_A(0, ra0, oa0), VDB_STEP_A(1, ra1, oa1))
            VDB_GROUP(1, fbB, faB, fbA, faA, VDB_STEP_A(2, ra2, oa2), VDB_STEP_A(3, ra3, oa3))
            VDB_GROUP(2, fbA, faA, fbB, faB, VDB_STEP_B(0, rb0), VDB_STEP_B(1, rb1))
            VDB_GROUP(3, fbB, faB, fbA, faA, VDB_STEP_B(2, rb2), VDB_STEP_B(3, rb3))
#undef VDB_GROUP
        } else {
            VDB_STEP_C()
            VDB_STEP_A(0, ra0, oa0) VDB_STEP_A(1, ra1, oa1) VDB_STEP_A(2, ra2, oa2) VDB_STEP_A(3, ra3, oa3)
            VDB_STEP_B(0, rb0) VDB_STEP_B(1, rb1) VDB_STEP_B(2, rb2) VDB_STEP_B(3, rb3)
            // only the valid 32-row blocks (wave-uniform predicate)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 fb = *reinterpret_cast<const float4*>(bptr + g * 32);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (rp * MT + i < mt_valid) {
                        const float4 fa = *reinterpret_cast<const float4*>(aptr + i * 32 * ROWB + g * 32);
                        acc[i] = VDB_MFMA(fa.x, fb.x, acc[i]);
                        acc[i] = VDB_MFMA(fa.y, fb.y, acc[i]);
                        acc[i] = VDB_MFMA(fa.z, fb.z, acc[i]);
                        acc[i] = VDB_MFMA(fa.w, fb.w, acc[i]);
                    }
                }
            }
        }
#undef VDB_STEP_A
#undef VDB_STEP_B
#undef VDB_STEP_C
        if (!(p.ablate & 4u)) __syncthreads();

        if (ks == KS - 1 && !(p.ablate & 8u)) {
            // ---- epilogue of this tile: ranking score, inclusive threshold, append the rare survivors
            // to this lane's private sub-pool (no atomics, no cross-lane traffic)
            const uint32_t par = tile % 3;
            const float* al = sAlpha + par * TR + 4 * h;
            const float* be = sBeta + par * TR + 4 * h;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t mtg = rp * MT + i;
                if (!PARTIAL || mtg < mt_valid) {
                    const uint32_t vbits = sValid[par * (TR / 32) + mtg] >> (4 * h);
                    const uint32_t rowb = tr0 + mtg * 32 + 4 * h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 a4 = *reinterpret_cast<const float4*>(al + mtg * 32 + 8 * j);
                        const float4 b4 = *reinterpret_cast<const float4*>(be + mtg * 32 + 8 * j);
#define VDB_PUSH(E, AC, BC)                                                                        \
    {                                                                                              \
        const float sc_ = fmaf(acc[i][4 * j + (E)], (AC), (BC));                                   \
        if (!(sc_ > thrq)) {                                                                       \
            const bool ok_ = (vbits >> (8 * j + (E))) & 1u;                                        \
            if (pcnt < p.capl) mypool[pcnt] = make_raw_key(sc_, ok_ ? rowb + 8 * j + (E) : 0xffffffffu); \
            ++pcnt;                                                                                \
        }                                                                                          \
    }
                        VDB_PUSH(0, a4.x, b4.x)
                        VDB_PUSH(1, a4.y, b4.y)
                        VDB_PUSH(2, a4.z, b4.z)
                        VDB_PUSH(3, a4.w, b4.w)
#undef VDB_PUSH
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        }
        tile = tile1; ks = ks1;
        tile1 = tile2; ks1 = ks2;
    };

    const uint32_t full_tiles = (r1 - r0) / TR;                         // tiles with all TR rows valid
    const uint32_t full_stages = full_tiles * KS;
    uint32_t st = 0;
    for (; st < full_stages; ++st) run_stage(st, std::false_type{});
    for (; st < total; ++st) run_stage(st, std::true_type{});
    p.pool_cnt[sub] = pcnt;                                             // may exceed capl: the select flags it
#undef VDB_LA
#undef VDB_TILE_OFFSETS
#undef VDB_LB
#undef VDB_LC
#undef VDB_SA
#undef VDB_SB
#undef VDB_SC
}

// nqt = 32-query tiles per workgroup actually needed (1..4); n_super = workgroups along the query axis
void launch_fused(const FusedParams& p, int nqt, uint32_t n_super, hipStream_t s) {
    dim3 grid(p.n_wg, n_super);
    size_t lds = fused_lds_bytes(nqt);
    switch (nqt) {
    case 1: hipLaunchKernelGGL((fused_score_filter_kernel<1, 1, 4>), grid, dim3(256), lds, s, p); break;
    case 2: hipLaunchKernelGGL((fused_score_filter_kernel<2, 2, 4>), grid, dim3(256), lds, s, p); break;
#ifdef VDB_DIAG
    case 8: hipLaunchKernelGGL((fused_score_filter_kernel<8, 4, 8>), grid, dim3(512), lds, s, p); break;   // register-staged 256-query shape (A/B against kernels_fused_dma3.hip)
#endif
    default: hipLaunchKernelGGL((fused_score_filter_kernel<4, 4, 4>), grid, dim3(256), lds, s, p); break;
    }
}
// sub-pools per query for a launch with n_wg row ranges: [n_wg][RP][2]
uint32_t fused_subpools_per_query(int nqt, uint32_t n_wg) {
    int rp = nqt == 1 ? CfgQ32::RP : nqt == 2 ? CfgQ64::RP : nqt == 8 ? CfgQ256::RP : CfgQ128::RP;
    return n_wg * rp * 2;
}

}  // namespace vdb
