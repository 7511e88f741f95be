// kernels_fused_s16.hip -- the FILTER pass of the screening tier over the bf16 SHADOW of the rows (vdb_flat_set_shadow):
// the same scores as kernels_fused_bf16p.hip, bit for bit (the shadow holds exactly the v_cvt_pk_bf16_f32 roundings that
// kernel makes in registers, and every accumulator sees the same MFMAs in the same order), from half the HBM bytes.
//
// Half the bytes halve the time the row stream leaves for everything else, so the shape is built around the matrix pipe
// and not around the stream any more:
//   * FOUR waves per workgroup, one per SIMD, each with the full 512-register file: a wave owns 128 rows x 128 queries
//     (4 x 4 MFMA tiles of 32 x 32, 256 accumulator registers) instead of 128 x 64 -- one LDS fragment byte now feeds
//     1.33x the MFMAs (8 fragment reads per 16 MFMAs instead of 6 per 8), and the f32 -> bf16 conversions are gone;
//   * rows are fetched 128 B at a time (K = 64 bf16: one full line per request), queries 64 B at a time from the image
//     query_prep already writes per K = 32 step.  The unit of the loop is the HALF-STAGE (K = 32: two MFMA k-steps):
//       row ring    3 x 32 KB, indexed by stage = half-stage >> 1, refilled on odd half-stages (two stages in flight)
//       query ring  3 x 16 KB, indexed by half-stage, refilled every half-stage (from L2)
//   * ONE barrier per half-stage, in its middle, exactly as in the pipelined f32-row kernel: k-step 0 runs with the
//     fragment reads of k-step 1 between its MFMAs; counted wait + barrier publish half-stage u+1 and free the images of
//     u; k-step 1 runs with the fragment reads of u+1 and the DMA of u+3 between its MFMAs.
// vmcnt accounting (in-order completion): a wave issues per half-stage v, after its barrier, [4 row constants if a tile
// starts], 4 query pieces of v+3 and, when v is odd, 8 row pieces of stage (v>>1)+3.  At the wait of half-stage u the
// queries of u+1 -- issued at v = u-2 -- must have landed: whatever was issued after them is 8 + 4 pieces in either
// parity (u even: Q(u+2) and R of v = u-1; u odd: R of v = u-2 and Q(u+2)), so the wait is vmcnt(12); the rows of the
// next stage are older than that.  The half-stage after a tile start allows 4 more (the constants sit in the window).
#include "kernels.h"

#include <type_traits>

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NW = 4, NT = NW * 64;
constexpr int TR = 256;                          // rows per tile
constexpr int TQ = 256;                          // queries per tile
constexpr int A_ROWB = 128;                      // 64 bf16 per row and row stage
constexpr int B_ROWB = 64;                       // 32 bf16 per query and half-stage
constexpr int A_BYTES = TR * A_ROWB;             // 32 KB
constexpr int B_BYTES = TQ * B_ROWB;             // 16 KB
constexpr int R_OFF = 0, Q_OFF = 3 * A_BYTES, IMG_BYTES = 3 * A_BYTES + 3 * B_BYTES;   // 144 KB
constexpr int MT = 4, QT = 4;                    // MFMA tiles per wave: 4 x 32 rows, 4 x 32 queries

#ifdef VDB_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The 256 accumulators of a wave live in a[0:255], named BY HAND.  Left to hipcc, a 4 x 4 block of f32x16 accumulators
// that fills the AccVGPR file exactly is allocated with copies through VGPRs and scratch spills inside the stage loop (a
// scratch reload is a vector-memory operation: it drains the DMA pipeline).  So the MFMAs, the zeroing and the epilogue's
// reads are inline asm on fixed AccVGPRs -- accumulator (i, j) is a[16 (4 i + j) .. +15] -- and the compiler allocates
// nothing but the 256 architectural VGPRs.  Inline asm is opaque to the hazard recognizer: the waits between an MFMA
// and the first VALU access to its result are placed by hand (s_nop before the epilogue and after the zeroing).
#define VDB_ALL_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
#define VDB_MFMA(I, J, FA, FB)                                                                         \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]"                               \
                 :: "v"(FA), "v"(FB), "n"(16 * (4 * (I) + (J))), "n"(16 * (4 * (I) + (J)) + 15) : VDB_ALL_AGPRS)
// (the AccVGPR clobber list on a READ keeps hipcc from parking values in "free" AccVGPRs across it -- it has no other way to
// know that a[0:255] are taken; the Makefile checks the generated code for compiler-made AccVGPR accesses and scratch)
#define VDB_ACC_READ(DST, I, J, R) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(DST) : "n"(16 * (4 * (I) + (J)) + (R)) : VDB_ALL_AGPRS)
#define VDB_Z1(N) asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(N) : VDB_ALL_AGPRS);
#define VDB_Z4(N) VDB_Z1(N) VDB_Z1((N) + 1) VDB_Z1((N) + 2) VDB_Z1((N) + 3)
#define VDB_Z16(N) VDB_Z4(N) VDB_Z4((N) + 4) VDB_Z4((N) + 8) VDB_Z4((N) + 12)
#define VDB_Z64(N) VDB_Z16(N) VDB_Z16((N) + 16) VDB_Z16((N) + 32) VDB_Z16((N) + 48)
#define VDB_ZERO_ACC { VDB_Z64(0) VDB_Z64(64) VDB_Z64(128) VDB_Z64(192) asm volatile("s_nop 7" ::: "memory"); }
}  // namespace

// SAMPLE = true: the sample pass over the COMPACT bf16 copy of the sample rows (p.rows16 = that copy, sample j at row j: the
// rows the f32 sample pass gathers with 3 KB strides, here contiguous and half the bytes).  One tile per workgroup; the
// epilogue keeps the smallest eligible score per (tile, row half, lane half) and query instead of filtering -- the same
// groups, the same scores and therefore the same thresholds as kernels_fused_bf16.hip's sample mode.
template <bool SAMPLE, bool MARGIN>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void fused_s16_kernel(FusedBf16Params p) {
    __shared__ __attribute__((aligned(16))) char smem[IMG_BYTES];
    __shared__ __attribute__((aligned(16))) float sAlpha[2 * TR];
    __shared__ __attribute__((aligned(16))) float sBeta[2 * TR];
    __shared__ __attribute__((aligned(16))) uint32_t sMaskW[2 * TR];
    __shared__ __attribute__((aligned(16))) float sMarg[MARGIN ? 2 * TR : 4];
    __shared__ float sG[MARGIN ? TQ : 1];

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wr = w >> 1, wq = w & 1;                             // row half, query half of this wave
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t ld = p.ld;                                           // elements; a multiple of 64 (host checks)
    const uint32_t KH = ld / 32;                                        // half-stages per tile (even)
    const uint32_t rowb = ld * 2;                                       // bytes per shadow row

    // ---- the rows this workgroup covers: whole tiles dealt round-robin, as in the f32-row kernel (same sub-pools, same keys)
    const uint32_t nblk = ((SAMPLE ? p.n_sample : p.n_rows) + TR - 1) / TR;
    const uint32_t r0 = blockIdx.x * TR;
    const uint32_t r1 = SAMPLE ? p.n_sample : p.n_rows;
    const uint32_t ntiles = SAMPLE ? 1u : (blockIdx.x < nblk ? (nblk - blockIdx.x + p.n_wg - 1) / p.n_wg : 0);
    const uint32_t TS = SAMPLE ? TR : p.n_wg * TR;                      // rows between consecutive tiles of this workgroup
    // sample index -> device row: positions spread evenly over the rows, consecutive positions in different tiles (see
    // kernels_fused_bf16.hip)
    auto sample_row_of = [&](uint32_t j) -> uint32_t {
        const uint32_t pos = (j & 255u) * (p.n_sample >> 8) + (j >> 8);
        return (uint32_t)(((uint64_t)pos * p.n_rows) >> p.sample_shift);
    };

    uint32_t q_of[QT];
    uint64_t* pool[QT];
    size_t sub[QT];
    float thr[QT];
    uint32_t pcnt[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        q_of[j] = wq * 128 + 32 * j + c;
        sub[j] = 0; pool[j] = nullptr; thr[j] = 0.f; pcnt[j] = 0;
        if (SAMPLE) continue;
        sub[j] = (((size_t)blockIdx.x * TQ + q_of[j]) * 2 + wr) * 2 + h;      // counts workgroup-major, like the keys
        pool[j] = p.pool + ((((size_t)blockIdx.x * TQ + q_of[j]) * 2 + wr) * 2 + h) * p.capl;
        thr[j] = p.thr[q_of[j]];
        if (kDiag && (p.ablate & 16u)) thr[j] = -__builtin_inff();
        asm volatile("" : "+v"(thr[j]));                                // consumed here: no ordinary load pending in the loop
    }
    if (ntiles == 0) {
#pragma unroll
        for (int j = 0; j < QT; ++j) p.pool_cnt[sub[j]] = 0;
        return;
    }
    if (SAMPLE && r0 >= r1) return;
    const bool no_nan = !SAMPLE && fused_no_nan(p.scalars, p.qmax_bits, !MARGIN);     // wave-uniform: can a score of this launch be NaN?
    const uint32_t NS = ntiles * (KH / 2);                              // row stages (= pairs of half-stages) of this workgroup
    const uint32_t last_row = p.n_rows - 1;
    const char* __restrict__ rows_b = reinterpret_cast<const char*>(p.rows16);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qb);

    // ---- DMA plan.  Row image: 32 pieces of 1 KB (8 rows x 128 B); wave w fills pieces 8w..8w+7 (lane L -> row L>>3,
    // 16-byte position L&7).  Query image: 16 pieces of 1 KB, wave w fills 4w..4w+3, contiguous in global memory.
    // Row image swizzle: data chunk x of row r at position x ^ ((r>>1)&7); query image: x ^ ((r>>2)&3) (query_prep).
    // With ONE wave per SIMD the wave's own instruction stream is the limit (one instruction per 4 cycles; an MFMA lasts
    // 32), so a piece costs three instructions: addresses are a uniform 64-bit base in SGPRs, advanced once per stage,
    // plus a per-lane 32-bit offset that never changes, and the M0 write rides in front of an MFMA (which is also the
    // wait state M0 needs before the DMA instruction).
    const uint32_t a_pr = lane >> 3, a_pp = lane & 7;
    const uint32_t a_chunk0 = (a_pp ^ ((a_pr >> 1) & 7)) * 16, a_chunk1 = (a_pp ^ ((4 + (a_pr >> 1)) & 7)) * 16;
    uint32_t voffR[8], voffQ[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) voffR[j] = (8 * j + a_pr) * rowb + ((j & 1) ? a_chunk1 : a_chunk0);
#pragma unroll
    for (int j = 0; j < 4; ++j) voffQ[j] = j * 1024 + lane * 16;
    const uint32_t KS64 = KH / 2;
    // fetch state: next row stage (tile ft, K position fk in units of 128 B) and next query half-stage (image fq).  Past the
    // last stage the fetch keeps running over the last tile (the counted waits need every look-ahead to exist; those
    // images are never read, their lines are L2 hits)
    uint32_t ft = 0, fk = 0, fq = 0;
    const char* baseR = rows_b + (size_t)(r0 + 64 * w) * rowb;
    const char* baseQ = bbase + (4 * w) * 1024;
    const size_t tile_jump = (size_t)TS * rowb - (size_t)(KS64 - 1) * 128;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const uint32_t slotR0 = lds0 + R_OFF + (8 * w) * 1024, slotQ0 = lds0 + Q_OFF + (4 * w) * 1024;
    auto adv_rows = [&]() {
        ++fk;
        if (fk == KS64) {
            fk = 0;
            if (ft + 1 < ntiles) { ++ft; baseR += tile_jump; } else baseR -= (size_t)(KS64 - 1) * 128;
        } else baseR += 128;
    };
    auto adv_query = [&]() {
        ++fq;
        if (fq == KH) { fq = 0; baseQ -= (size_t)(KH - 1) * B_BYTES; } else baseQ += B_BYTES;
    };
#define VDB_DMA_SV(VOFF, BASE, M0V, NTS)                                                               \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" NTS                  \
                 :: "v"(VOFF), "s"(BASE), "s"(M0V) : "memory", "m0")
#define VDB_MFMA_DMA(I, J, FA, FB, VOFF, BASE, SLOT, PIECE, NTS)                                       \
    asm volatile("s_add_u32 m0, %6, %7\n\tv_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]\n\tglobal_load_lds_dwordx4 %4, %5" NTS \
                 :: "v"(FA), "v"(FB), "n"(16 * (4 * (I) + (J))), "n"(16 * (4 * (I) + (J)) + 15), "v"(VOFF), "s"(BASE), "s"(SLOT), \
                    "n"((PIECE) * 1024) : "memory", "m0", "scc", VDB_ALL_AGPRS)
#define VDB_MFMA4(I, FA, FB)                                                                           \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%5:%6], %0, %1, a[%5:%6]\n\tv_mfma_f32_32x32x16_bf16 a[%7:%8], %0, %2, a[%7:%8]\n\t" \
                 "v_mfma_f32_32x32x16_bf16 a[%9:%10], %0, %3, a[%9:%10]\n\tv_mfma_f32_32x32x16_bf16 a[%11:%12], %0, %4, a[%11:%12]" \
                 :: "v"(FA), "v"(FB[0]), "v"(FB[1]), "v"(FB[2]), "v"(FB[3]),                          \
                    "n"(64 * (I)), "n"(64 * (I) + 15), "n"(64 * (I) + 16), "n"(64 * (I) + 31), "n"(64 * (I) + 32), "n"(64 * (I) + 47), \
                    "n"(64 * (I) + 48), "n"(64 * (I) + 63) : VDB_ALL_AGPRS)
#define VDB_DMA4(GP, LP)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"                        \
                 :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)(LP)), "v"((const void*)(GP)) : "memory", "m0")
    // row constants of a tile, one tile ahead (4 bytes per lane): wave w fetches those of tile rows 64w..64w+63
    auto issue_consts = [&](uint32_t t) {
        const uint32_t par = t & 1u;
        uint32_t row = r0 + t * TS + 64 * w + lane;
        if (SAMPLE) row = sample_row_of(row < p.n_sample ? row : p.n_sample - 1);     // (row was the sample index)
        row = row > last_row ? last_row : row;
        VDB_DMA4(p.alpha + row, sAlpha + par * TR + 64 * w);
        VDB_DMA4(p.beta + row, sBeta + par * TR + 64 * w);
        VDB_DMA4(p.rowmask + (row >> 5), sMaskW + par * TR + 64 * w);
        if (MARGIN) VDB_DMA4(p.margin + row, sMarg + par * TR + 64 * w);
    };
    if (MARGIN) {
        if (tid < TQ) { float g = p.qg[tid]; asm volatile("" : "+v"(g)); sG[tid] = g; }
    }

    VDB_ZERO_ACC

    // fragment read offsets.  Row fragment of k-step t of half hf: chunk 4 hf + 2 t + h; query fragment: chunk 2 t + h
    const uint32_t swa = (c >> 1) & 7, swb = (c >> 2) & 3;
    const uint32_t a_base = (wr * 128 + c) * A_ROWB;                    // + i*32*A_ROWB
    const uint32_t b_base = Q_OFF + (wq * 128 + c) * B_ROWB;            // + j*32*B_ROWB
    uint32_t offA[2][2], offB[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < 2; ++t) offA[hf][t] = a_base + ((4 * hf + 2 * t + h) ^ swa) * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) offB[t] = b_base + ((2 * t + h) ^ swb) * 16;

    // ---- prologue: constants of tile 0, then R0 Q0 R1 Q1 Q2 R2 (the steady-state order from there on)
    issue_consts(0);
#define VDB_ISSUE_R(SLOT) { _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) VDB_DMA_SV(voffR[j_], baseR, slotR0 + (SLOT) * A_BYTES + j_ * 1024, " nt"); adv_rows(); }
#define VDB_ISSUE_Q(SLOT) { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) VDB_DMA_SV(voffQ[j_], baseQ, slotQ0 + (SLOT) * B_BYTES + j_ * 1024, ""); adv_query(); }
    VDB_ISSUE_R(0) VDB_ISSUE_Q(0) VDB_ISSUE_R(1) VDB_ISSUE_Q(1) VDB_ISSUE_Q(2) VDB_ISSUE_R(2)
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                   // R0 and Q0 landed (R1 Q1 Q2 R2 may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 fa0[MT], fb0[QT];                                            // k-step 0 fragments of the half-stage computed next
#define VDB_RD(OFF) (*reinterpret_cast<const bf16x8*>(smem + (OFF)))
#pragma unroll
    for (int i = 0; i < MT; ++i) fa0[i] = VDB_RD(offA[0][0] + i * 32 * A_ROWB);
#pragma unroll
    for (int j = 0; j < QT; ++j) fb0[j] = VDB_RD(offB[0] + j * 32 * B_ROWB);

    uint32_t tile = 0, ks = 0;                                          // of the half-stage being computed (ks in half-stages)
    uint32_t rcur = 0, qcur = 0;                                        // byte offsets of its row / query image in their rings
    bool consts_in_window = false;                                      // the previous half-stage issued row constants

    auto run_half = [&](auto hf_tag) {
        constexpr int HF = decltype(hf_tag)::value;
        bf16x8 fa1[MT], fb1[QT];
        // images of the next half-stage
        const uint32_t rn = HF == 0 ? rcur : (rcur == 2 * A_BYTES ? 0u : rcur + A_BYTES);
        const uint32_t qn = qcur == 2 * B_BYTES ? 0u : qcur + B_BYTES;
        // ---- k-step 0: 16 MFMAs, the eight fragment reads of k-step 1 between them
#pragma unroll
        for (int j = 0; j < QT; ++j) fb1[j] = VDB_RD(qcur + offB[1] + j * 32 * B_ROWB);
#define VDB_K0_ROW(I)                                                                                  \
        fa1[I] = VDB_RD(rcur + offA[HF][1] + (I) * 32 * A_ROWB);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        VDB_MFMA4(I, fa0[I], fb0);                                                                     \
        __builtin_amdgcn_sched_barrier(0);
        VDB_K0_ROW(0) VDB_K0_ROW(1) VDB_K0_ROW(2) VDB_K0_ROW(3)
#undef VDB_K0_ROW
        // ---- publish the next half-stage, free the images of this one
        if (consts_in_window) {                                         // + the 3 (4) constant fetches of the tile start
            if (MARGIN) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15) lgkmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        consts_in_window = false;
        if (ks == 0 && tile + 1 < ntiles) { issue_consts(tile + 1); consts_in_window = true; }
        // ---- k-step 1: 16 MFMAs; with them the fragment reads of k-step 0 of the next half-stage, the 4 query pieces of
        // half-stage +3 (into the query image this half-stage just finished with) and, on odd half-stages, the 8 row pieces
        // of stage +3 (into the row image).  ORDER: the query pieces first -- the counted waits rely on every query piece of
        // a half-stage being older than its row pieces
#pragma unroll
        for (int j = 0; j < QT; ++j) fb0[j] = VDB_RD(qn + offB[0] + j * 32 * B_ROWB);
        const uint32_t mq = slotQ0 + qcur, mr = slotR0 + rcur;
        if (kDiag && (p.ablate & 2u)) {                                  // diagnostics: the stage loop without any DMA (compute only)
            fa0[0] = VDB_RD(rn + offA[HF ^ 1][0] + 0 * 32 * A_ROWB); __builtin_amdgcn_sched_barrier(0); VDB_MFMA4(0, fa1[0], fb1); __builtin_amdgcn_sched_barrier(0);
            fa0[1] = VDB_RD(rn + offA[HF ^ 1][0] + 1 * 32 * A_ROWB); __builtin_amdgcn_sched_barrier(0); VDB_MFMA4(1, fa1[1], fb1); __builtin_amdgcn_sched_barrier(0);
            fa0[2] = VDB_RD(rn + offA[HF ^ 1][0] + 2 * 32 * A_ROWB); __builtin_amdgcn_sched_barrier(0); VDB_MFMA4(2, fa1[2], fb1); __builtin_amdgcn_sched_barrier(0);
            fa0[3] = VDB_RD(rn + offA[HF ^ 1][0] + 3 * 32 * A_ROWB); __builtin_amdgcn_sched_barrier(0); VDB_MFMA4(3, fa1[3], fb1); __builtin_amdgcn_sched_barrier(0);
        } else {
        fa0[0] = VDB_RD(rn + offA[HF ^ 1][0] + 0 * 32 * A_ROWB);
        __builtin_amdgcn_sched_barrier(0);
        VDB_MFMA_DMA(0, 0, fa1[0], fb1[0], voffQ[0], baseQ, mq, 0, "");
        VDB_MFMA_DMA(0, 1, fa1[0], fb1[1], voffQ[1], baseQ, mq, 1, "");
        VDB_MFMA_DMA(0, 2, fa1[0], fb1[2], voffQ[2], baseQ, mq, 2, "");
        VDB_MFMA_DMA(0, 3, fa1[0], fb1[3], voffQ[3], baseQ, mq, 3, "");
        __builtin_amdgcn_sched_barrier(0);
        fa0[1] = VDB_RD(rn + offA[HF ^ 1][0] + 1 * 32 * A_ROWB);
        __builtin_amdgcn_sched_barrier(0);
        if (HF == 1) {
            VDB_MFMA_DMA(1, 0, fa1[1], fb1[0], voffR[0], baseR, mr, 0, " nt");
            VDB_MFMA_DMA(1, 1, fa1[1], fb1[1], voffR[1], baseR, mr, 1, " nt");
            VDB_MFMA_DMA(1, 2, fa1[1], fb1[2], voffR[2], baseR, mr, 2, " nt");
            VDB_MFMA_DMA(1, 3, fa1[1], fb1[3], voffR[3], baseR, mr, 3, " nt");
        } else VDB_MFMA4(1, fa1[1], fb1);
        __builtin_amdgcn_sched_barrier(0);
        fa0[2] = VDB_RD(rn + offA[HF ^ 1][0] + 2 * 32 * A_ROWB);
        __builtin_amdgcn_sched_barrier(0);
        if (HF == 1) {
            VDB_MFMA_DMA(2, 0, fa1[2], fb1[0], voffR[4], baseR, mr, 4, " nt");
            VDB_MFMA_DMA(2, 1, fa1[2], fb1[1], voffR[5], baseR, mr, 5, " nt");
            VDB_MFMA_DMA(2, 2, fa1[2], fb1[2], voffR[6], baseR, mr, 6, " nt");
            VDB_MFMA_DMA(2, 3, fa1[2], fb1[3], voffR[7], baseR, mr, 7, " nt");
        } else VDB_MFMA4(2, fa1[2], fb1);
        __builtin_amdgcn_sched_barrier(0);
        fa0[3] = VDB_RD(rn + offA[HF ^ 1][0] + 3 * 32 * A_ROWB);
        __builtin_amdgcn_sched_barrier(0);
        VDB_MFMA4(3, fa1[3], fb1);
        __builtin_amdgcn_sched_barrier(0);
        }
        adv_query();
        if (HF == 1) adv_rows();
        rcur = rn; qcur = qn;

        if (HF == 1 && ks == KH - 1 && !(kDiag && (p.ablate & 8u))) {   // (KH is even: a tile ends on an odd half-stage)
            const uint32_t par = tile & 1u;
            const uint32_t tr0 = r0 + tile * TS;
            // eligibility of this wave's 128 rows: two ballots over (in range) & (mask bit of the row)
            unsigned long long val[2] = {0ull, 0ull};
            if (SAMPLE) {
                // eligibility folded into beta: thread t owns tile row t; an ineligible row gets beta = +inf, so its scores are
                // +inf (or NaN) and never a group minimum -- no per-element select in the scoring below
                const uint32_t rt = tid;
                const uint32_t sj = tr0 + rt < r1 ? tr0 + rt : r1 - 1;
                const bool ok = tr0 + rt < r1 && ((sMaskW[par * TR + rt] >> (sample_row_of(sj) & 31)) & 1u);
                if (!ok) sBeta[par * TR + rt] = __uint_as_float(0x7f800000u);
                __syncthreads();
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint32_t rt = wr * 128 + 64 * m + lane;
                    val[m] = __ballot(tr0 + rt < r1 && ((sMaskW[par * TR + rt] >> (rt & 31)) & 1u));       // (tr0 is a multiple of 32)
                }
            }
            const float* al = sAlpha + par * TR + wr * 128 + 4 * h;
            const float* be = sBeta + par * TR + wr * 128 + 4 * h;
            const float* mg = sMarg + (MARGIN ? par * TR + wr * 128 : 0);
            // MARGIN: common path on the plain score against a per-tile loosened threshold, exact lower-bound test on the rare
            // path (see kernels_fused_bf16p.hip)
            float thp[QT], ng[QT];
#pragma unroll
            for (int j = 0; j < QT; ++j) { thp[j] = thr[j]; ng[j] = 0.f; }
            if (MARGIN) {
                float mm = fmaxf(mg[2 * lane], mg[2 * lane + 1]);
                for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
#pragma unroll
                for (int j = 0; j < QT; ++j) {
                    const float g = sG[q_of[j]];
                    ng[j] = -g;
                    thp[j] = fmaf(g, mm, thr[j]);
                    thp[j] += (fabsf(thr[j]) + g * mm) * 6.0e-7f;
                }
            }
            float best[QT];                                             // sample mode: running group minima
#pragma unroll
            for (int j = 0; j < QT; ++j) best[j] = __uint_as_float(0x7f800000u);
            asm volatile("s_nop 7" ::: "memory");                       // the last MFMA of the k-step wrote a[240:255]; they are read last
#define VDB_EPI_J(I, G4, J, NN)                                                                        \
    {                                                                                                  \
        float p0_, p1_, p2_, p3_;                              /* four reads in ONE asm: hipcc puts an s_nop behind every inline asm */ \
        asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]" \
                     : "=v"(p0_), "=v"(p1_), "=v"(p2_), "=v"(p3_)                                      \
                     : "n"(16 * (4 * (I) + (J)) + 4 * (G4)), "n"(16 * (4 * (I) + (J)) + 4 * (G4) + 1), \
                       "n"(16 * (4 * (I) + (J)) + 4 * (G4) + 2), "n"(16 * (4 * (I) + (J)) + 4 * (G4) + 3) : VDB_ALL_AGPRS); \
        const f32x2 p01 = {p0_, p1_}, p23 = {p2_, p3_};                                                \
        const f32x2 s01 = __builtin_elementwise_fma(p01, al01, be01), s23 = __builtin_elementwise_fma(p23, al23, be23); \
        const float s0 = s01.x, s1 = s01.y, s2 = s23.x, s3 = s23.y;                                    \
        if (SAMPLE) {                                                                                  \
            /* smallest score of the lane's rows; ineligible rows score +inf (their beta was replaced above), and v_min_f32 */ \
            /* skips a NaN score: such a row is no witness for a threshold */                          \
            best[J] = fminf(fminf(best[J], s0), s1); best[J] = fminf(fminf(best[J], s2), s3);          \
            asm volatile("" : "+v"(best[J]));      /* computed HERE: left alone, hipcc sinks the min chains of three of the four */ \
                                                   /* queries to the end of the epilogue and keeps 192 scores alive (spills) */ \
        } else {                                                                                       \
        const float tp = thp[J];                                                                       \
        /* ONE compare for the four rows: their smallest score against the threshold; where a score of the launch could */ \
        /* be NaN (fused_no_nan says no) a NaN-propagating sum is tested as well (kernels_fused_bf16p.hip) */ \
        const f32x2 mn_ = __builtin_elementwise_min(s01, s23);                                         \
        unsigned long long m = __builtin_amdgcn_ballot_w64(!(fminf(mn_.x, mn_.y) > tp));                \
        if (!(NN)) {                                           /* NN: compile-time copy of no_nan (the epilogue exists twice) */ \
            const f32x2 u_ = s01 + s23; const float t_ = u_.x + u_.y;                                  \
            m |= __builtin_amdgcn_ballot_w64(t_ != t_);                                                \
        }                                                                                              \
        if (__builtin_expect(m != 0ull, 0)) {                                                          \
            uint32_t hm = (!(s0 > tp) ? 1u : 0u) | (!(s1 > tp) ? 2u : 0u) | (!(s2 > tp) ? 4u : 0u) | (!(s3 > tp) ? 8u : 0u); \
            hm &= (vbits >> (8 * (G4))) & 0xfu;                                                        \
            while (hm) {                                                                               \
                const uint32_t e = (uint32_t)__builtin_ctz(hm);                                        \
                hm &= hm - 1u;                                                                         \
                float sc = e == 0 ? s0 : e == 1 ? s1 : e == 2 ? s2 : s3;                               \
                if (MARGIN) {                                                                          \
                    sc = fmaf(ng[J], mg[(I) * 32 + 8 * (G4) + 4 * h + e], sc);                         \
                    if (sc > thr[J]) continue;                                                         \
                }                                                                                      \
                if (!(kDiag && (p.ablate & 32u)) && pcnt[J] < p.capl) pool[J][pcnt[J]] = make_raw_key(sc, tr0 + rt0 + e); \
                ++pcnt[J];                                                                             \
            }                                                                                          \
        }                                                                                              \
        }                                                                                              \
    }
#define VDB_EPI_G(I, G4, NN)                                                                           \
    {                                                                                                  \
        const float4 a4 = *reinterpret_cast<const float4*>(al + (I) * 32 + 8 * (G4));                  \
        const float4 b4 = *reinterpret_cast<const float4*>(be + (I) * 32 + 8 * (G4));                  \
        const f32x2 al01 = {a4.x, a4.y}, al23 = {a4.z, a4.w}, be01 = {b4.x, b4.y}, be23 = {b4.z, b4.w}; \
        const uint32_t rt0 = rowt + 8 * (G4);                                                          \
        VDB_EPI_J(I, G4, 0, NN) __builtin_amdgcn_sched_barrier(0); VDB_EPI_J(I, G4, 1, NN) __builtin_amdgcn_sched_barrier(0);    \
        VDB_EPI_J(I, G4, 2, NN) __builtin_amdgcn_sched_barrier(0); VDB_EPI_J(I, G4, 3, NN) __builtin_amdgcn_sched_barrier(0);    \
    }
#define VDB_EPI_I(I, NN)                                                                               \
    {                                                                                                  \
        const uint32_t vbits = (uint32_t)(val[(I) >> 1] >> (32 * ((I) & 1) + 4 * h));                  \
        const uint32_t rowt = wr * 128 + (I) * 32 + 4 * h;              /* tile-row of element (g = 0, e = 0) */ \
        VDB_EPI_G(I, 0, NN) VDB_EPI_G(I, 1, NN) VDB_EPI_G(I, 2, NN) VDB_EPI_G(I, 3, NN)                \
    }
            // two copies of the scoring code, chosen once per tile: with the NaN test per group and without it
            if (no_nan) { VDB_EPI_I(0, 1) VDB_EPI_I(1, 1) VDB_EPI_I(2, 1) VDB_EPI_I(3, 1) }
            else { VDB_EPI_I(0, 0) VDB_EPI_I(1, 0) VDB_EPI_I(2, 0) VDB_EPI_I(3, 0) }
#undef VDB_EPI_I
#undef VDB_EPI_G
#undef VDB_EPI_J
            if (SAMPLE) {
                // one group minimum per (tile, row half, lane half) and query; the key's low word only has to make the keys of
                // one query distinct: the group index
                const uint32_t g = ((blockIdx.x * 2 + wr) * 2 + h);
#pragma unroll
                for (int j = 0; j < QT; ++j)
                    p.minkeys[(size_t)q_of[j] * p.minkey_stride + g] = best[j] < __uint_as_float(0x7f800000u) ? make_key(best[j], g) : EMPTY_KEY;
            }
            VDB_ZERO_ACC
        }
        ++ks;
        if (ks == KH) { ks = 0; ++tile; }
    };

    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    for (uint32_t s = 0; s < NS; ++s) {
        run_half(H0{});
        run_half(H1{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the look-ahead fetches past the last stage target this workgroup's LDS
    if (!SAMPLE) {
#pragma unroll
        for (int j = 0; j < QT; ++j) p.pool_cnt[sub[j]] = pcnt[j];
    }
#undef VDB_DMA_SV
#undef VDB_MFMA_DMA
#undef VDB_MFMA4
#undef VDB_DMA4
#undef VDB_ISSUE_R
#undef VDB_ISSUE_Q
#undef VDB_RD
}

void launch_fused_s16(const FusedBf16Params& p, hipStream_t s) {
    if (p.margin) hipLaunchKernelGGL((fused_s16_kernel<false, true>), dim3(p.n_wg), dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((fused_s16_kernel<false, false>), dim3(p.n_wg), dim3(NT), 0, s, p);
}

// the sample pass over the compact bf16 sample copy (p.rows16): plain scores, as launch_sample_bf16
void launch_sample_s16(const FusedBf16Params& p, hipStream_t s) {
    const uint32_t stiles = (p.n_sample + TR - 1) / TR;
    if (!stiles) return;
    FusedBf16Params q = p;
    q.margin = nullptr; q.qg = nullptr;
    hipLaunchKernelGGL((fused_s16_kernel<true, false>), dim3(stiles), dim3(NT), 0, s, q);
}

}  // namespace vdb
