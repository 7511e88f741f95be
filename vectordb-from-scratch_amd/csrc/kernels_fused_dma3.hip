// kernels_fused_dma3.hip -- three-image ring version of kernels_fused_dma.hip: the per-stage barrier sits in the
// MIDDLE of a stage (it publishes stage s+1, whose LDS-DMA was issued a full stage earlier), so a wave runs
// from the last MFMA group of stage s straight into stage s+1 (whose first fragments it has already read)
// instead of draining the MFMA pipe at an end-of-stage barrier.
// kernels_fused_dma.hip -- variant of the fused score+filter kernel whose K stages are brought into LDS
// by LDS-DMA (`global_load_lds_dwordx4`: global memory -> LDS with no VGPR round trip and no ds_write),
// for the headline shape only (8 waves, 128 rows x 256 queries).  Same MFMA sequence, K order, score
// expression and private candidate pools as kernels_fused.hip, so results are bit-identical.
//
// LDS image: rows of 128 B, unpadded (an LDS-DMA wave instruction writes 1 KB = 8 rows linearly, so
// padding is impossible); bank conflicts are avoided by an XOR swizzle applied on BOTH sides:
//   16-byte chunk x of row r lives at chunk position x ^ ((r >> 1) & 7)
//   - DMA side: the lane that fills position c' of row r fetches data chunk c' ^ ((r >> 1) & 7)
//   - read side: ds_read_b128 of data chunk x reads position x ^ ((r >> 1) & 7)
// 16 lanes of a ds_read_b128 group hold 16 rows distinct mod 16 -> 16 distinct 16-byte slots of the
// 256-byte bank row (2 rows per bank row x 8 permuted chunks).
#include "kernels.h"

#include <type_traits>

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define VDB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int NQT = 8, MT = 4, NW = 8;
constexpr int NT = NW * 64;
constexpr int TR = 32 * MT;                      // 128 rows per tile (RP = 1)
constexpr int ROWB = 128;                        // unpadded
constexpr int A_BYTES = TR * ROWB;               // 16 KB
constexpr int B_BYTES = 32 * NQT * ROWB;         // 32 KB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // 48 KB



typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
}  // namespace


__global__ __launch_bounds__(NT, 2) void fused_score_filter_dma3_kernel(FusedParams p) {
    // The two stage images are DISTINCT LDS objects and every access names its image at compile time
    // (the stage loop is unrolled by two): hipcc's waitcnt pass can then tell that a ds_read of one image
    // does not alias the LDS-DMA in flight into the other, and does not put a vmcnt(0) in front of it.
    __shared__ __attribute__((aligned(16))) char sImg0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg1[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char sImg2[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) float sAlpha[3 * TR];
    __shared__ __attribute__((aligned(16))) float sBeta[3 * TR];
    __shared__ uint32_t sValid[3 * (TR / 32)];

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t qt = w;                                              // one 32-query tile per wave

    const uint32_t nblk = (p.n_rows + 31) >> 5;
    const uint32_t b0 = (uint32_t)(((uint64_t)blockIdx.x * nblk) / p.n_wg);
    const uint32_t b1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * nblk) / p.n_wg);
    const uint32_t r0 = b0 * 32;
    const uint32_t r1 = (b1 * 32 < p.n_rows) ? b1 * 32 : p.n_rows;
    const uint32_t qwg = p.q_base + blockIdx.y * (32 * NQT);
    const uint32_t q = qwg + qt * 32 + c;
    const size_t sub = ((size_t)q * p.n_wg + blockIdx.x) * 2 + h;
    uint64_t* mypool = p.pool + sub * p.capl;
    uint32_t pcnt = 0;
    if (r0 >= r1) {
        p.pool_cnt[sub] = 0;
        return;
    }
    const uint32_t ntiles = (r1 - r0 + TR - 1) / TR;
    const uint32_t KS = p.ld / KSTAGE;
    const uint32_t total = ntiles * KS;
    const float thrq = p.thr[q];
    const uint32_t ld = p.ld;
    const uint32_t last_row = p.n_rows - 1;
    const uint32_t rows_wg = last_row - r0;
    const char* __restrict__ abase = reinterpret_cast<const char*>(p.rows + (size_t)r0 * ld);
    const char* __restrict__ bbase = reinterpret_cast<const char*>(p.qp + (size_t)qwg * ld);

    // ---- DMA plan: a stage is 48 pieces of 1 KB (8 rows x 128 B).  Wave w fills A pieces 2w, 2w+1 and B
    // pieces 4w..4w+3.  Lane L fills chunk position L&7 of row L>>3 of the piece.
    const uint32_t prr = lane >> 3, pcp = lane & 7;
    const uint32_t arow0 = 8 * (2 * w) + prr, arow1 = arow0 + 8;        // rows inside the 128-row tile
    const uint32_t achk0 = (pcp ^ ((arow0 >> 1) & 7)) * 16, achk1 = (pcp ^ ((arow1 >> 1) & 7)) * 16;
    uint32_t oa0 = 0, oa1 = 0;                                          // per-tile byte offsets (clamped rows)
    uint32_t ob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t brow = 8 * (4 * w + j) + prr;                    // query row inside the 256-query tile
        ob[j] = brow * ld * 4 + (pcp ^ ((brow >> 1) & 7)) * 16;
    }
    const uint32_t ob0 = ob[0], ob1 = ob[1], ob2 = ob[2], ob3 = ob[3];
    uint32_t oc_row = 0;
    bool r_inrange = false, r_inrange_c = false;
    float r_alpha = 0.f, r_beta = 0.f;
    uint32_t r_mask = 0, r_bit = 0;

#define VDB_TILE_OFFSETS(TILE)                                                                         \
    {                                                                                                  \
        uint32_t q0_ = (TILE) * TR + arow0, q1_ = (TILE) * TR + arow1;                                 \
        q0_ = q0_ > rows_wg ? rows_wg : q0_; q1_ = q1_ > rows_wg ? rows_wg : q1_;                      \
        oa0 = q0_ * ld * 4 + achk0; oa1 = q1_ * ld * 4 + achk1;                                        \
        const uint32_t cr_ = (TILE) * TR + (tid % TR);                                                 \
        r_inrange = r0 + cr_ < r1;                                                                     \
        oc_row = r_inrange ? cr_ : rows_wg;                                                            \
    }
#define VDB_DMA(GP, IMG, LOFF) __builtin_amdgcn_global_load_lds((glb_ptr_t)(GP), (lds_ptr_t)((IMG) + (LOFF)), 16, 0, 0)
    // all 6 pieces of this wave for k-stage KSI of the tile whose offsets are current, into LDS image BUF
#define VDB_ISSUE(IMG, KSI)                                                                            \
    {                                                                                                  \
        const uint32_t la_ = (2 * w) * 1024;                                                           \
        const uint32_t lb_ = A_BYTES + (4 * w) * 1024;                                                 \
        const uint32_t ko_ = (KSI) * (KSTAGE * 4);                                                     \
        VDB_DMA(abase + (oa0 + ko_), IMG, la_);                                                        \
        VDB_DMA(abase + (oa1 + ko_), IMG, la_ + 1024);                                                 \
        VDB_DMA(bbase + (ob0 + ko_), IMG, lb_);                                                        \
        VDB_DMA(bbase + (ob1 + ko_), IMG, lb_ + 1024);                                                 \
        VDB_DMA(bbase + (ob2 + ko_), IMG, lb_ + 2048);                                                 \
        VDB_DMA(bbase + (ob3 + ko_), IMG, lb_ + 3072);                                                 \
    }
#define VDB_LC()                                                                                       \
    {                                                                                                  \
        const uint32_t rr_ = r0 + oc_row;                                                              \
        r_inrange_c = r_inrange; /* the prefetch below may move the offsets on to the next tile */     \
        r_bit = rr_ & 31;                                                                              \
        r_mask = p.rowmask[rr_ >> 5];                                                                  \
        r_alpha = p.alpha[rr_];                                                                        \
        r_beta = p.beta[rr_];                                                                          \
    }
#define VDB_SC(TILE)                                                                                   \
    {                                                                                                  \
        const uint32_t par_ = (TILE) % 3;                                                              \
        const bool ok_ = r_inrange_c && ((r_mask >> r_bit) & 1u);                                        \
        sAlpha[par_ * TR + (tid % TR)] = r_alpha;                                                      \
        sBeta[par_ * TR + (tid % TR)] = ok_ ? r_beta : __uint_as_float(0x7f800000u);                   \
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

    // swizzled fragment read offsets: data chunk (2g+h) of row (..+c) sits at position (2g+h) ^ ((c>>1)&7)
    const uint32_t sw = (c >> 1) & 7;
    const uint32_t rd0 = ((0 + h) ^ sw) * 16, rd1 = ((2 + h) ^ sw) * 16, rd2 = ((4 + h) ^ sw) * 16, rd3 = ((6 + h) ^ sw) * 16;
    const uint32_t a_row_off = c * ROWB;                                // + i*32*ROWB
    const uint32_t b_row_off = A_BYTES + (qt * 32 + c) * ROWB;

    // ---- prologue: stages 0 and 1 into images 0 and 1, published by one barrier
    uint32_t tile = 0, ks = 0;
    uint32_t otile = 0;                                                 // tile the DMA offsets currently describe
    VDB_TILE_OFFSETS(0u)
    VDB_ISSUE(sImg0, 0u)
    if (total > 1) {
        const uint32_t t1 = (KS == 1) ? 1u : 0u, k1 = (KS == 1) ? 0u : 1u;
        if (t1 != otile) { VDB_TILE_OFFSETS(t1) otile = t1; }
        VDB_ISSUE(sImg1, k1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // fragments of K group 0 of the stage about to run are always read one stage ahead ("carried")
    float4 fbC, faC[MT];
    fbC = *reinterpret_cast<const float4*>(sImg0 + b_row_off + rd0);
#pragma unroll
    for (int i = 0; i < MT; ++i) faC[i] = *reinterpret_cast<const float4*>(sImg0 + a_row_off + i * 32 * ROWB + rd0);

    auto run_stage = [&](uint32_t st, auto partial_tag, auto buf_tag) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        constexpr int BUF = decltype(buf_tag)::value;                   // LDS image this stage computes from
        const char* img = BUF == 0 ? sImg0 : BUF == 1 ? sImg1 : sImg2;
        const char* img_next = BUF == 0 ? sImg1 : BUF == 1 ? sImg2 : sImg0;     // stage st+1 (already landed)
        char* img_fill = BUF == 0 ? sImg2 : BUF == 1 ? sImg0 : sImg1;           // stage st+2 goes here
        const uint32_t tr0 = r0 + tile * TR;
        const uint32_t mt_valid = PARTIAL ? ((r1 - tr0 + 31) >> 5) : (uint32_t)(TR / 32);
        uint32_t tile1 = tile, ks1 = ks + 1;
        if (ks1 == KS) { ks1 = 0; ++tile1; }
        uint32_t tile2 = tile1, ks2 = ks1 + 1;
        if (ks2 == KS) { ks2 = 0; ++tile2; }
        if (ks == 0) {                                                  // first stage of a tile: fetch its row constants
            const uint32_t cr_ = tile * TR + (tid % TR);
            r_inrange_c = r0 + cr_ < r1;
            const uint32_t rr_ = r0 + (r_inrange_c ? cr_ : rows_wg);
            r_bit = rr_ & 31;
            r_mask = p.rowmask[rr_ >> 5];
            r_alpha = p.alpha[rr_];
            r_beta = p.beta[rr_];
        }
        const char* ap = img + a_row_off;
        const char* bp = img + b_row_off;
        float4 fbB, faB[MT];
#define VDB_MFMA4(FA, FB, COMP)                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) if (!PARTIAL || (uint32_t)i < mt_valid) acc[i] = VDB_MFMA(FA[i].COMP, FB.COMP, acc[i]);
#define VDB_READ(FB, FA, BASEA, BASEB, RD)                                                             \
    FB = *reinterpret_cast<const float4*>((BASEB) + (RD));                                             \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) FA[i] = *reinterpret_cast<const float4*>((BASEA) + i * 32 * ROWB + (RD));
        // ---- K groups 0 and 1
        VDB_MFMA4(faC, fbC, x) VDB_MFMA4(faC, fbC, y)
        VDB_READ(fbB, faB, ap, bp, rd1)
        VDB_MFMA4(faC, fbC, z) VDB_MFMA4(faC, fbC, w)
        VDB_MFMA4(faB, fbB, x) VDB_MFMA4(faB, fbB, y)
        VDB_READ(fbC, faC, ap, bp, rd2)
        VDB_MFMA4(faB, fbB, z) VDB_MFMA4(faB, fbB, w)
        // ---- mid-stage: publish stage st+1 (its DMA was issued one stage ago), then start stage st+2's DMA
        if (ks == 0) { VDB_SC(tile) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 2 < total) {
            if (otile != tile2) { VDB_TILE_OFFSETS(tile2) otile = tile2; }
            VDB_ISSUE(img_fill, ks2)
        }
        // ---- K groups 2 and 3; the last one also reads group 0 of the next stage from its (published) image
        VDB_MFMA4(faC, fbC, x) VDB_MFMA4(faC, fbC, y)
        VDB_READ(fbB, faB, ap, bp, rd3)
        VDB_MFMA4(faC, fbC, z) VDB_MFMA4(faC, fbC, w)
        VDB_MFMA4(faB, fbB, x) VDB_MFMA4(faB, fbB, y)
        if (st + 1 < total) { VDB_READ(fbC, faC, img_next + a_row_off, img_next + b_row_off, rd0) }
        VDB_MFMA4(faB, fbB, z) VDB_MFMA4(faB, fbB, w)
#undef VDB_MFMA4
#undef VDB_READ

        if (ks == KS - 1) {
            const uint32_t par = tile % 3;
            const float* al = sAlpha + par * TR + 4 * h;
            const float* be = sBeta + par * TR + 4 * h;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t mtg = i;
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
        tile = tile1;
        ks = ks1;
    };

    const uint32_t full_tiles = (r1 - r0) / TR;
    const uint32_t full_stages = full_tiles * KS;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    uint32_t st = 0;
    for (; st + 2 < full_stages; st += 3) {                             // stage index mod 3 == image index
        run_stage(st, std::false_type{}, B0{});
        run_stage(st + 1, std::false_type{}, B1{});
        run_stage(st + 2, std::false_type{}, B2{});
    }
    for (; st < total; ++st) {
        const uint32_t m = st % 3;
        if (st < full_stages) {
            if (m == 0) run_stage(st, std::false_type{}, B0{}); else if (m == 1) run_stage(st, std::false_type{}, B1{}); else run_stage(st, std::false_type{}, B2{});
        } else {
            if (m == 0) run_stage(st, std::true_type{}, B0{}); else if (m == 1) run_stage(st, std::true_type{}, B1{}); else run_stage(st, std::true_type{}, B2{});
        }
    }
    p.pool_cnt[sub] = pcnt;
#undef VDB_TILE_OFFSETS
#undef VDB_DMA
#undef VDB_ISSUE
#undef VDB_LC
#undef VDB_SC
}

void launch_fused_dma3(const FusedParams& p, uint32_t n_super, hipStream_t s) {
    hipLaunchKernelGGL(fused_score_filter_dma3_kernel, dim3(p.n_wg, n_super), dim3(NT), 0, s, p);
}

}  // namespace vdb
