// kernels_fused.hip -- the hot kernel: batched query x database inner products on
// v_mfma_f32_32x32x2_f32 (exact f32 in / f32 accumulate), fused with the ranking-score
// epilogue and an inclusive threshold filter that appends the few surviving
// (score, row) keys to a per-query candidate pool.  The B x N score matrix never
// exists in memory.   gfx950 (MI355X, CDNA4) only.
//
// Workgroup = 512 threads = 8 waves (2 per SIMD), one workgroup per CU.
// Tile      = TR database rows x (32*NQT) queries, K staged 32 floats at a time.
//   wave w: query tile qt = w % NQT, row part rp = w / NQT, MT accumulator tiles of
//           32x32 (rows (rp*MT+i)*32.., i < MT).  Headline shape NQT = 8, MT = 4:
//           128 rows x 256 queries, 64 accumulator VGPRs per lane, 113 KB LDS.
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

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROWB = 144;                 // padded LDS bytes per 32-float row

// NQT = 32-query tiles per workgroup (1,2,4,8); MT = 32-row accumulator tiles per wave.
// RP = 8/NQT waves share a query tile and split the tile's rows; TR = 32*MT*RP rows per tile.
template <int NQT, int MT> struct FusedCfg {
    static constexpr int RP = 8 / NQT;
    static constexpr int TR = 32 * MT * RP;
    static constexpr int A_BYTES = TR * ROWB;
    static constexpr int B_BYTES = 32 * NQT * ROWB;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int CONST_OFF = 2 * STAGE_BYTES;           // alpha[3][TR] beta[3][TR] valid[3][TR/32]
    static constexpr int LDS_BYTES = CONST_OFF + 3 * TR * 4 * 2 + 3 * (TR / 32) * 4;
    static constexpr int NA = (TR * 8) / 512;                   // A float4 loads per thread per stage
    static constexpr int NB = (NQT * 256 + 511) / 512;          // B float4 loads per thread per stage
};
// the four shipped shapes: (queries per workgroup, rows per tile)
using Cfg8 = FusedCfg<8, 4>;   // 256 queries x 128 rows, 64 accumulator VGPRs per lane
using Cfg4 = FusedCfg<4, 4>;   // 128 queries x 256 rows
using Cfg2 = FusedCfg<2, 2>;   //  64 queries x 256 rows
using Cfg1 = FusedCfg<1, 1>;   //  32 queries x 256 rows

size_t fused_lds_bytes(int nqt) {
    switch (nqt) {
    case 1: return Cfg1::LDS_BYTES;
    case 2: return Cfg2::LDS_BYTES;
    case 4: return Cfg4::LDS_BYTES;
    default: return Cfg8::LDS_BYTES;
    }
}
uint32_t fused_tile_rows(int nqt) {
    switch (nqt) {
    case 1: return Cfg1::TR;
    case 2: return Cfg2::TR;
    case 4: return Cfg4::TR;
    default: return Cfg8::TR;
    }
}

#define VDB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NQT, int MT>
__global__ __launch_bounds__(512, 2) void fused_score_filter_kernel(FusedParams p) {
    using C = FusedCfg<NQT, MT>;
    constexpr int TR = C::TR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per-tile row constants, 3 rotating slots: slot (t+1)%3 is written during tile t's last stage while
    // slot t%3 is read by tile t's epilogue; with 3 slots this also holds when a tile is a single stage
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
    if (r0 >= r1) return;
    const uint32_t ntiles = (r1 - r0 + TR - 1) / TR;
    const uint32_t KS = p.ld / KSTAGE;
    const uint32_t total = ntiles * KS;
    const uint32_t qwg = p.q_base + blockIdx.y * (32 * NQT);            // first query of this workgroup
    const uint32_t q = qwg + qt * 32 + c;                              // this lane's query
    const float thrq = p.thr[q];
    const float* __restrict__ grow = p.rows;
    const float* __restrict__ gq = p.qp + (size_t)qwg * p.ld;
    const uint32_t ld = p.ld;

    // ---- staging registers (global -> VGPR -> LDS); named scalars, not arrays, so they stay in VGPRs
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float r_alpha = 0.f, r_beta = 0.f;
    bool r_valid = false;

#define VDB_LA(I, REG)                                                                                          \
    if constexpr (C::NA > (I)) {                                                                                \
        uint32_t idx_ = (I) * 512 + tid;                                                                        \
        uint32_t row_ = tr0_ + (idx_ >> 3);                                                                     \
        if (row_ >= p.n_rows) row_ = p.n_rows - 1; /* in bounds; masked in the epilogue */                      \
        REG = *reinterpret_cast<const float4*>(grow + (size_t)row_ * ld + ks_ * KSTAGE + (idx_ & 7) * 4);       \
    }
#define VDB_LB(I, REG)                                                                                          \
    if constexpr (C::NB > (I)) {                                                                                \
        uint32_t idx_ = (I) * 512 + tid;                                                                        \
        if (idx_ < NQT * 256)                                                                                   \
            REG = *reinterpret_cast<const float4*>(gq + (size_t)(idx_ >> 3) * ld + ks_ * KSTAGE + (idx_ & 7) * 4); \
    }
#define ISSUE_LOADS(ST)                                                                                         \
    {                                                                                                           \
        const uint32_t tile_ = (ST) / KS, ks_ = (ST) - tile_ * KS;                                              \
        const uint32_t tr0_ = r0 + tile_ * TR;                                                                  \
        VDB_LA(0, ra0) VDB_LA(1, ra1) VDB_LA(2, ra2) VDB_LA(3, ra3)                                             \
        VDB_LB(0, rb0) VDB_LB(1, rb1) VDB_LB(2, rb2) VDB_LB(3, rb3)                                             \
        if (ks_ == 0 && tid < TR) {                                                                             \
            uint32_t row_ = tr0_ + tid;                                                                         \
            bool ok_ = row_ < r1;                                                                               \
            uint32_t rr_ = ok_ ? row_ : p.n_rows - 1;                                                           \
            if (ok_ && p.rowmask) ok_ = (p.rowmask[rr_ >> 5] >> (rr_ & 31)) & 1u;                               \
            r_valid = ok_;                                                                                      \
            r_alpha = p.alpha[rr_];                                                                             \
            r_beta = ok_ ? p.beta[rr_] : __uint_as_float(0x7f800000u); /* +inf never passes a finite thr */     \
        }                                                                                                       \
    }

#define VDB_SA(I, REG)                                                                                          \
    if constexpr (C::NA > (I)) {                                                                                \
        uint32_t idx_ = (I) * 512 + tid;                                                                        \
        *reinterpret_cast<float4*>(sa_ + (idx_ >> 3) * ROWB + (idx_ & 7) * 16) = REG;                           \
    }
#define VDB_SB(I, REG)                                                                                          \
    if constexpr (C::NB > (I)) {                                                                                \
        uint32_t idx_ = (I) * 512 + tid;                                                                        \
        if (idx_ < NQT * 256) *reinterpret_cast<float4*>(sb_ + (idx_ >> 3) * ROWB + (idx_ & 7) * 16) = REG;     \
    }
#define WRITE_LDS(ST)                                                                                           \
    {                                                                                                           \
        const uint32_t tile_ = (ST) / KS, ks_ = (ST) - tile_ * KS;                                              \
        char* sa_ = smem + ((ST) & 1) * C::STAGE_BYTES;                                                         \
        char* sb_ = sa_ + C::A_BYTES;                                                                           \
        VDB_SA(0, ra0) VDB_SA(1, ra1) VDB_SA(2, ra2) VDB_SA(3, ra3)                                             \
        VDB_SB(0, rb0) VDB_SB(1, rb1) VDB_SB(2, rb2) VDB_SB(3, rb3)                                             \
        if (ks_ == 0 && tid < TR) {                                                                             \
            const uint32_t par_ = tile_ % 3;                                                                    \
            sAlpha[par_ * TR + tid] = r_alpha;                                                                  \
            sBeta[par_ * TR + tid] = r_beta;                                                                    \
            unsigned long long bal_ = __ballot(r_valid); /* TR is a multiple of 64: whole waves */              \
            if (lane == 0) {                                                                                    \
                sValid[par_ * (TR / 32) + 2 * w] = (uint32_t)bal_;                                              \
                sValid[par_ * (TR / 32) + 2 * w + 1] = (uint32_t)(bal_ >> 32);                                  \
            }                                                                                                   \
        }                                                                                                       \
    }

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // ---- prologue
    ISSUE_LOADS(0u);
    WRITE_LDS(0u);
    __syncthreads();

    for (uint32_t st = 0; st < total; ++st) {
        const uint32_t tile = st / KS, ks = st - tile * KS;
        const uint32_t tr0 = r0 + tile * TR;
        const uint32_t mt_valid = (r1 - tr0 + 31) >> 5;                 // valid 32-row blocks left (may exceed the tile)
        const bool more = st + 1 < total;
        if (more) ISSUE_LOADS(st + 1);

        // ---- MFMAs of this stage
        const char* sa = smem + (st & 1) * C::STAGE_BYTES;
        const char* bptr = sa + C::A_BYTES + (qt * 32 + c) * ROWB + h * 16;
        const char* aptr = sa + (rp * MT * 32 + c) * ROWB + h * 16;
        if (mt_valid >= TR / 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 fb = *reinterpret_cast<const float4*>(bptr + g * 32);
                float4 fa[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const float4*>(aptr + i * 32 * ROWB + g * 32);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(fa[i].x, fb.x, acc[i]);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(fa[i].y, fb.y, acc[i]);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(fa[i].z, fb.z, acc[i]);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = VDB_MFMA(fa[i].w, fb.w, acc[i]);
            }
        } else {
            // partial last tile: only the valid 32-row blocks (wave-uniform predicate)
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

        if (more) WRITE_LDS(st + 1);
        __syncthreads();

        if (ks == KS - 1) {
            // ---- epilogue of this tile: score, inclusive threshold, rare append
            const uint32_t par = tile % 3;
            const float* al = sAlpha + par * TR + 4 * h;
            const float* be = sBeta + par * TR + 4 * h;
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint32_t mtg = rp * MT + i;
                if (mtg < mt_valid) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 a4 = *reinterpret_cast<const float4*>(al + mtg * 32 + 8 * j);
                        const float4 b4 = *reinterpret_cast<const float4*>(be + mtg * 32 + 8 * j);
                        cnt += !(fmaf(acc[i][4 * j + 0], a4.x, b4.x) > thrq);
                        cnt += !(fmaf(acc[i][4 * j + 1], a4.y, b4.y) > thrq);
                        cnt += !(fmaf(acc[i][4 * j + 2], a4.z, b4.z) > thrq);
                        cnt += !(fmaf(acc[i][4 * j + 3], a4.w, b4.w) > thrq);
                    }
                }
            }
            const uint32_t other = __shfl_xor(cnt, 32);
            const uint32_t tot = cnt + other;
            if (__ballot(tot != 0) != 0ull) {
                uint32_t base = 0;
                if (h == 0 && tot) base = atomicAdd(&p.pool_cnt[q], tot);
                base = __shfl(base, c);
                uint32_t off = base + (h ? other : 0u);
                uint64_t* pool = p.pool + (size_t)q * p.capq;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint32_t mtg = rp * MT + i;
                    if (mtg < mt_valid) {
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
            if (off < p.capq) pool[off] = ok_ ? make_key(sc_, rowb + 8 * j + (E)) : EMPTY_KEY;     \
            ++off;                                                                                 \
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
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        }
    }
#undef ISSUE_LOADS
#undef WRITE_LDS
#undef VDB_LA
#undef VDB_LB
#undef VDB_SA
#undef VDB_SB
}

void launch_fused(const FusedParams& p, int nqt, uint32_t n_super, hipStream_t s) {
    dim3 grid(p.n_wg, n_super), block(512);
    size_t lds = fused_lds_bytes(nqt);
    switch (nqt) {
    case 1: hipLaunchKernelGGL((fused_score_filter_kernel<1, 1>), grid, block, lds, s, p); break;
    case 2: hipLaunchKernelGGL((fused_score_filter_kernel<2, 2>), grid, block, lds, s, p); break;
    case 4: hipLaunchKernelGGL((fused_score_filter_kernel<4, 4>), grid, block, lds, s, p); break;
    default: hipLaunchKernelGGL((fused_score_filter_kernel<8, 4>), grid, block, lds, s, p); break;
    }
}

}  // namespace vdb
