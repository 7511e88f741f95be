// kernels_aux.hip -- everything around the fused MFMA kernel: row statistics, query
// preparation, dense sample scores (MFMA, no LDS), per-query radix select, exact
// re-rank + certification, exact scan fallback, multi-GPU partial merge.
// gfx950 only.  Built with -ffp-contract=off: the "exact" functions below must round
// every multiply and add separately, like the reference's scalar Rust
// (src/distance.rs:37-73, src/vector.rs:35-37).
#include "kernels.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace vdb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Exact-order arithmetic: sequential f32 left folds, one rounding per operation.
// sqrt uses __builtin_sqrtf (correctly rounded expansion); HIP's __fsqrt_rn lowers to a bare
// v_sqrt_f32 (1 ulp) on gfx950 and is NOT usable for bit parity.
// ---------------------------------------------------------------------------------------------
// Each fold loads 16 elements (4 x 16 bytes) before it consumes them, so the loads of a block are in
// flight together while the adds stay one strictly sequential chain.
__device__ __forceinline__ float fold_sq(const float* __restrict__ x, uint32_t d) {
    // vector.rs:35-37   sum_i x_i*x_i
    float s = 0.0f;
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = __fadd_rn(s, __fmul_rn(v[u].x, v[u].x));
            s = __fadd_rn(s, __fmul_rn(v[u].y, v[u].y));
            s = __fadd_rn(s, __fmul_rn(v[u].z, v[u].z));
            s = __fadd_rn(s, __fmul_rn(v[u].w, v[u].w));
        }
    }
    for (; i < d; ++i) s = __fadd_rn(s, __fmul_rn(x[i], x[i]));
    return s;
}

__device__ __forceinline__ float fold_dot(const float* __restrict__ q, const float* __restrict__ x, uint32_t d) {
    // distance.rs:67-73   sum_i a_i*b_i
    float s = 0.0f;
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(q + i + 4 * u);
            b[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = __fadd_rn(s, __fmul_rn(a[u].x, b[u].x));
            s = __fadd_rn(s, __fmul_rn(a[u].y, b[u].y));
            s = __fadd_rn(s, __fmul_rn(a[u].z, b[u].z));
            s = __fadd_rn(s, __fmul_rn(a[u].w, b[u].w));
        }
    }
    for (; i < d; ++i) s = __fadd_rn(s, __fmul_rn(q[i], x[i]));
    return s;
}

__device__ __forceinline__ float fold_sqdiff(const float* __restrict__ q, const float* __restrict__ x, uint32_t d) {
    // distance.rs:37-44   sum_i (a_i-b_i)^2   (powi(2) == t*t)
    float s = 0.0f;
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(q + i + 4 * u);
            b[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t;
            t = __fsub_rn(a[u].x, b[u].x); s = __fadd_rn(s, __fmul_rn(t, t));
            t = __fsub_rn(a[u].y, b[u].y); s = __fadd_rn(s, __fmul_rn(t, t));
            t = __fsub_rn(a[u].z, b[u].z); s = __fadd_rn(s, __fmul_rn(t, t));
            t = __fsub_rn(a[u].w, b[u].w); s = __fadd_rn(s, __fmul_rn(t, t));
        }
    }
    for (; i < d; ++i) { float t = __fsub_rn(q[i], x[i]); s = __fadd_rn(s, __fmul_rn(t, t)); }
    return s;
}

// The same folds over a SLICE of the pair, resumed from a partial sum: fold(q, x, d) == part(q + d1, x + d1, d - d1, part(q, x, d1, 0))
// for any d1 that is a multiple of 16 (the same sequence of roundings, element by element) -- rerank_kernel's K slices.
typedef float vdb_f2 __attribute__((ext_vector_type(2)));
// (products and differences two at a time -- v_pk_mul_f32 / v_pk_add_f32, each lane of a packed op rounds exactly like the
// scalar op -- the ADDS stay one by one, in order: the fold is the reference's.  One wave folds 48 candidates and is bound by
// its own instruction stream, so fewer instructions per element is time.)
__device__ __forceinline__ float fold_dot_part(const float* __restrict__ q, const float* __restrict__ x, uint32_t d, float s) {
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(q + i + 4 * u);
            b[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const vdb_f2 p01 = vdb_f2{a[u].x, a[u].y} * vdb_f2{b[u].x, b[u].y};
            const vdb_f2 p23 = vdb_f2{a[u].z, a[u].w} * vdb_f2{b[u].z, b[u].w};
            s = __fadd_rn(s, p01.x);
            s = __fadd_rn(s, p01.y);
            s = __fadd_rn(s, p23.x);
            s = __fadd_rn(s, p23.y);
        }
    }
    for (; i < d; ++i) s = __fadd_rn(s, __fmul_rn(q[i], x[i]));
    return s;
}
__device__ __forceinline__ float fold_sqdiff_part(const float* __restrict__ q, const float* __restrict__ x, uint32_t d, float s) {
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(q + i + 4 * u);
            b[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const vdb_f2 t01 = vdb_f2{a[u].x, a[u].y} - vdb_f2{b[u].x, b[u].y};
            const vdb_f2 t23 = vdb_f2{a[u].z, a[u].w} - vdb_f2{b[u].z, b[u].w};
            const vdb_f2 p01 = t01 * t01, p23 = t23 * t23;
            s = __fadd_rn(s, p01.x);
            s = __fadd_rn(s, p01.y);
            s = __fadd_rn(s, p23.x);
            s = __fadd_rn(s, p23.y);
        }
    }
    for (; i < d; ++i) { float t = __fsub_rn(q[i], x[i]); s = __fadd_rn(s, __fmul_rn(t, t)); }
    return s;
}
// the distance from the finished fold (s = sum of squared differences under Euclid, the dot product otherwise)
__device__ __forceinline__ float distance_from_fold(int metric, float s, float qn, float xn) {
    if (metric == EUCLID) return __builtin_sqrtf(s);
    if (metric == DOT) return -s;
    float den = __fmul_rn(qn, xn);                 // norm1 * norm2   distance.rs:58
    float sim = __fdiv_rn(s, den);
    if (sim < -1.0f) sim = -1.0f;                  // f32::clamp keeps NaN
    if (sim > 1.0f) sim = 1.0f;
    return __fsub_rn(1.0f, sim);
}

// DistanceMetric::distance (distance.rs:20-33) for one (query, row) pair.
// qn / xn are the exact-order norms of query and row (only read under Cosine).
__device__ __forceinline__ float exact_distance(int metric, const float* __restrict__ q,
                                                const float* __restrict__ x, uint32_t d, float qn, float xn) {
    if (metric == EUCLID) return __builtin_sqrtf(fold_sqdiff(q, x, d));
    float dot = fold_dot(q, x, d);
    if (metric == DOT) return -dot;
    float den = __fmul_rn(qn, xn);                 // norm1 * norm2   distance.rs:58
    float sim = __fdiv_rn(dot, den);
    if (sim < -1.0f) sim = -1.0f;                  // f32::clamp keeps NaN
    if (sim > 1.0f) sim = 1.0f;
    return __fsub_rn(1.0f, sim);
}

// ---------------------------------------------------------------------------------------------
// Row statistics at upload: exact-order norm and the (alpha, beta) of the ranking score
//   score = fma(dot, alpha, beta):  Euclid  nd2 - 2 dot ; Cosine  -dot/|d| ; Dot  -dot
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_round(float v) {        // the value v_cvt_pk_bf16_f32 (RNE) keeps
    return __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v) << 16);
}
__global__ __launch_bounds__(256) void row_stats_kernel(RowStatsParams p) {
    uint32_t row = p.row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    float nd2 = 0.0f, e2 = 0.0f, rel2 = 0.0f;
    if (row < p.row_end) {
        const float* x = p.rows + (size_t)row * p.ld;
        nd2 = fold_sq(x, p.dim);
        float nd = __builtin_sqrtf(nd2);
        p.nd[row] = nd;
        // bf16 rounding error of the row (screening-tier certification): |x - bf16(x)|^2 and |x|^2 accumulated in f64 (each
        // term is exact in f64, the sum is good to 2^-53 dim) and rounded UP when stored as f32 -- an f32 sum would be off by
        // up to dim 2^-24 relative, more than any fixed slack once dim reaches the thousands
        double e2d = 0.0, n2d = 0.0;
        for (uint32_t i = 0; i < p.dim; ++i) {
            const double xv = (double)x[i], e = xv - (double)bf16_round(x[i]);
            e2d += e * e;
            n2d += xv * xv;
        }
        auto up = [](double v) { float f = (float)v; return ((double)f < v) ? __uint_as_float(__float_as_uint(f) + 1u) : f; };   // v >= 0
        e2 = up(e2d * 1.0000001);
        rel2 = n2d > 0.0 ? up(e2d / n2d * 1.0000001) : 0.0f;
        if (!(e2 == e2)) e2 = 0.0f;                    // NaN / inf rows are caught by the NaN status, not by this bound
        if (!(rel2 == rel2)) rel2 = 0.0f;
        float a, b;
        if (p.metric == EUCLID) {
            a = -2.0f;
            // beta = |x|^2 minus the row's own share of the f32-fold error budget, rounded DOWN (a smaller score is safe)
            const double bd = (double)nd2 * (1.0 - (double)p.beta_shrink);
            b = (float)bd;
            if ((double)b > bd) b = __uint_as_float(__float_as_uint(b) - 1u);
            if (!p.margin) b = nd2;
        }
        else if (p.metric == COSINE) { a = -__fdiv_rn(1.0f, nd); b = 0.0f; }
        else { a = -1.0f; b = 0.0f; }
        p.alpha[row] = a;
        p.beta[row] = b;
        if (p.margin) {
            // true norms in f64 (the f32 fold `nd` may be off by dim 2^-24 relative)
            const double ed = sqrt(e2d), ndd = sqrt(n2d);
            const double m1 = (double)p.m_e * ed + (double)p.m_n * ndd, m2 = (double)p.m_b * ndd;
            const double m = (m1 > m2 ? m1 : m2) * 1.000001;
            float mf = (float)m;                       // inf for a row whose norm overflows: its lower-bound score is -inf, it is always a candidate
            if ((double)mf < m) mf = __uint_as_float(__float_as_uint(mf) + 1u);
            p.margin[row] = mf;                        // NaN rows: NaN margin -> NaN score -> kept by the filter -> NaN distance reported
            if (mf > 0.0f && mf < __uint_as_float(0x7f800000u)) atomicMax(p.nd2max_bits + 4, ~__float_as_uint(mf));   // smallest positive margin
        }
    }
    // wave max of the (non-negative or NaN) bit patterns, one atomic per wave
    uint32_t bits = __float_as_uint(nd2) & 0x7fffffffu, be = __float_as_uint(e2), br = __float_as_uint(rel2);
    uint32_t bmin = (row < p.row_end && nd2 > 0.0f) ? __float_as_uint(nd2) : 0xffffffffu;
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(bits, o); bits = t > bits ? t : bits;
        t = __shfl_xor(bmin, o); bmin = t < bmin ? t : bmin;
        t = __shfl_xor(be, o); be = t > be ? t : be;
        t = __shfl_xor(br, o); br = t > br ? t : br;
    }
    if ((threadIdx.x & 63) == 0) {
        if (bits) atomicMax(p.nd2max_bits, bits);
        if (bmin != 0xffffffffu) atomicMax(p.nd2max_bits + 5, ~bmin);      // smallest positive fold(x*x), stored complemented (0 = none yet)
        if (be) atomicMax(p.nd2max_bits + 2, be);
        if (br) atomicMax(p.nd2max_bits + 3, br);
    }
}
void launch_row_stats(const RowStatsParams& p, hipStream_t s) {
    uint32_t n = p.row_end - p.row_begin;
    if (!n) return;
    hipLaunchKernelGGL(row_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
}

// bf16 shadow copy of the rows (vdb_flat_set_shadow): the same RNE conversion the f32-row screening kernels apply to
// their fragments, done once at upload; 8 elements per thread
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ rows, uint16_t* __restrict__ rows16,
                                                           size_t first8, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const float4* src = reinterpret_cast<const float4*>(rows) + 2 * (first8 + i);
    const float4 lo = src[0], hi = src[1];
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    auto pk = [](float x, float y) { f2 v = {x, y}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)); };
    uint4 o = {pk(lo.x, lo.y), pk(lo.z, lo.w), pk(hi.x, hi.y), pk(hi.z, hi.w)};
    reinterpret_cast<uint4*>(rows16)[first8 + i] = o;
}
void launch_rows_to_bf16(const float* rows, uint16_t* rows16, uint32_t ld, uint32_t row_begin, uint32_t row_end, hipStream_t s) {
    if (row_end <= row_begin) return;
    const size_t first8 = (size_t)row_begin * ld / 8, n8 = (size_t)(row_end - row_begin) * ld / 8;   // ld is a multiple of 32
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, rows, rows16, first8, n8);
}

__global__ __launch_bounds__(256) void sample_to_bf16_kernel(const float* __restrict__ rows, uint32_t ld8, uint32_t n_rows, uint32_t n_sample,
                                                             uint32_t shift, uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // 8 elements per thread
    if (i >= (size_t)n_sample * ld8) return;
    const uint32_t j = (uint32_t)(i / ld8), c8 = (uint32_t)(i % ld8);
    const uint32_t pos = (j & 255u) * (n_sample >> 8) + (j >> 8);
    const uint32_t row = (uint32_t)(((uint64_t)pos * n_rows) >> shift);
    const float4* src = reinterpret_cast<const float4*>(rows + (size_t)row * ld8 * 8) + 2 * c8;
    const float4 lo = src[0], hi = src[1];
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    auto pk = [](float x, float y) { f2 v = {x, y}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)); };
    uint4 o = {pk(lo.x, lo.y), pk(lo.z, lo.w), pk(hi.x, hi.y), pk(hi.z, hi.w)};
    reinterpret_cast<uint4*>(out)[i] = o;
}
void launch_sample_to_bf16(const float* rows, uint32_t ld, uint32_t n_rows, uint32_t n_sample, uint32_t shift, uint16_t* out, hipStream_t s) {
    if (!n_sample || !n_rows) return;
    const size_t n8 = (size_t)n_sample * (ld / 8);
    hipLaunchKernelGGL(sample_to_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, rows, ld / 8, n_rows, n_sample, shift, out);
}

__global__ __launch_bounds__(256) void count_zero_live_kernel(const float* nd, const uint32_t* livemask,
                                                              uint32_t n_rows, uint32_t* out) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    bool z = false;
    if (row < n_rows) {
        bool live = livemask ? ((livemask[row >> 5] >> (row & 31)) & 1u) : true;
        z = live && nd[row] == 0.0f;
    }
    unsigned long long b = __ballot(z);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (uint32_t)__popcll(b));
}
void launch_count_zero_live(const float* nd, const uint32_t* livemask, uint32_t n_rows, uint32_t* out,
                            hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(count_zero_live_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, nd, livemask,
                       n_rows, out);
}

__global__ __launch_bounds__(256) void build_rowmask_kernel(const uint64_t* row_ids, const uint32_t* livemask,
                                                            const uint64_t* idmask, uint64_t mask_bits,
                                                            uint32_t n_rows, uint32_t* out) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (row < n_rows) {
        bool live = livemask ? ((livemask[row >> 5] >> (row & 31)) & 1u) : true;
        uint64_t id = row_ids[row];
        ok = live && id < mask_bits && ((idmask[id >> 6] >> (id & 63)) & 1ull);
    }
    unsigned long long b = __ballot(ok);
    uint32_t lane = threadIdx.x & 63;
    uint32_t w0 = (blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) >> 5;
    uint32_t nwords = (n_rows + 31) >> 5;
    if (lane == 0 && w0 < nwords) out[w0] = (uint32_t)b;
    if (lane == 1 && w0 + 1 < nwords) out[w0 + 1] = (uint32_t)(b >> 32);
}
void launch_build_rowmask(const uint64_t* row_ids, const uint32_t* livemask, const uint64_t* idmask,
                          uint64_t mask_bits, uint32_t n_rows, uint32_t* out_mask, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(build_rowmask_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, row_ids, livemask,
                       idmask, mask_bits, n_rows, out_mask);
}

// ---------------------------------------------------------------------------------------------
// Query preparation: copy into the zero-padded [nq_pad][ld] block the MFMA kernels read,
// and compute each query's exact-order norm (one wave per query row; lane 0 folds).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void query_prep_kernel(QueryPrepParams p) {
    extern __shared__ __attribute__((aligned(16))) float sQrow[];      // [ld]
    uint32_t q = blockIdx.x;
    float* dst = p.qp + (size_t)q * p.ld;
    const float* src = p.q_in + (size_t)q * p.dim;
    for (uint32_t i = threadIdx.x; i < p.ld; i += blockDim.x) {
        float v = (q < p.nq && i < p.dim) ? src[i] : 0.0f;
        dst[i] = v;
        sQrow[i] = v;
        if (p.qb) {
            // bf16 copy (v_cvt_pk_bf16_f32: RNE, NaN stays NaN) in the LDS image order of the screening kernel: per pass of
            // 256 queries and per K stage of 32 one 16 KB image, query r = 64 B, its 16-byte chunk x at position x ^ ((r>>2)&3)
            const uint32_t pass = q >> 8, r = q & 255u, ks = i >> 5, x = (i >> 3) & 3u, e = i & 7u;
            const size_t img = ((size_t)pass * (p.ld >> 5) + ks) * (256u * 32u);
            p.qb[img + r * 32u + ((x ^ ((r >> 2) & 3u)) << 3) + e] = __builtin_bit_cast(uint16_t, (__bf16)v);
        }
    }
    if (p.qb) {
        // |q - bf16(q)|: any summation order will do (it is an upper bound, rounded up below)
        float e2 = 0.0f;
        for (uint32_t i = threadIdx.x; i < p.ld; i += blockDim.x) {
            float v = (q < p.nq && i < p.dim) ? src[i] : 0.0f;
            float e = v - bf16_round(v);
            e2 += e * e;
        }
        for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
        __shared__ float sE[4];
        if ((threadIdx.x & 63) == 0) sE[threadIdx.x >> 6] = e2;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = (sE[0] + sE[1]) + (sE[2] + sE[3]);
            // f32 partial sums of non-negative terms: relative error below (ld/256 + 8) 2^-24 per thread chain and tree; the
            // slack grows with the row length so that it holds up to the largest supported dimension
            t = __builtin_sqrtf(t) * (1.0001f + (float)p.ld * 1.0e-7f);
            p.qerr[q] = (t == t) ? t : 0.0f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float n = 0.0f;
        if (q < p.nq) {
            n = __builtin_sqrtf(fold_sq(sQrow, p.dim));                // sequential fold, from LDS
            if (p.metric == COSINE && n == 0.0f) atomicOr(p.status, ST_ZERO_QUERY);
            if (p.qb) atomicMax(p.status + 2, __float_as_uint(n));   // largest query norm of the search (NaN bits > inf bits > finite): fused_no_nan
        }
        p.qnorm[q] = n;
        if (p.qb && p.qg) {
            // g_q = (|q| + kappa |q - bf16(q)|), rounded up; |q| from the f32 fold may be low by ld 2^-24 relative
            const float g = (n * (1.00001f + (float)p.ld * 1.2e-7f) + p.kappa * p.qerr[q]) * 1.00001f;
            p.qg[q] = (q < p.nq) ? g : 0.0f;
        }
        if (q >= p.nq) p.thr[q] = __uint_as_float(0xff800000u);   // -inf
        if (q < p.nq && p.clear_a) { p.clear_a[q] = 0u; p.clear_b[q] = 0u; }
    }
}
void launch_query_prep(const QueryPrepParams& p, hipStream_t s) {
    hipLaunchKernelGGL(query_prep_kernel, dim3(p.nq_pad), dim3(256), (size_t)p.ld * sizeof(float), s, p);
}

// ---------------------------------------------------------------------------------------------
// Dense scores of a row sample (or of every row when the index is small).  One wave =
// one 32-row x 32-query tile on v_mfma_f32_32x32x2_f32, operands straight from global
// memory.  The K order (groups of 8: k = 8g+s from lanes 0-31, 8g+4+s from lanes 32-63,
// s = 0..3) and the score expression are EXACTLY those of the fused kernel, so the two
// produce bit-identical scores and the sample's kk-th score is a valid inclusive
// threshold there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sample_row(uint32_t j, uint32_t n_sample, uint32_t n_rows) {
    return n_sample >= n_rows ? j : (uint32_t)(((uint64_t)j * n_rows) / n_sample);
}

__global__ __launch_bounds__(256) void dense_scores_kernel(DenseParams p) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t c = lane & 31, h = lane >> 5;
    const uint32_t tile = blockIdx.x * 4 + w;
    const uint32_t ntiles = (p.n_sample + 31) / 32;
    if (tile >= ntiles) return;                       // whole wave exits together
    uint32_t sj = tile * 32 + c;
    if (sj >= p.n_sample) sj = p.n_sample - 1;
    const uint32_t srow = sample_row(sj, p.n_sample, p.n_rows);
    const float* ap = p.rows + (size_t)srow * p.ld + 4 * h;
    const float* bp = p.qp + (size_t)(blockIdx.y * 32 + c) * p.ld + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // ld is a multiple of 32, so the K groups of 8 come in blocks of 4.  Only ~1 wave per SIMD is
    // in flight here, so each wave keeps two blocks of operand loads ahead of its MFMAs.
    const uint32_t nblocks = p.ld / 32;
    float4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
#define VDB_DLOAD(A, B, BLK)                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        A[u] = *reinterpret_cast<const float4*>(ap + 32 * (BLK) + 8 * u);                     \
        B[u] = *reinterpret_cast<const float4*>(bp + 32 * (BLK) + 8 * u);                     \
    }
#define VDB_DMFMA(A, B)                                                                       \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u].x, B[u].x, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u].y, B[u].y, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u].z, B[u].z, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u].w, B[u].w, acc, 0, 0, 0);             \
    }
    VDB_DLOAD(a0, b0, 0u)
    if (nblocks > 1) { VDB_DLOAD(a1, b1, 1u) }
    for (uint32_t blk = 0; blk < nblocks; blk += 3) {
        if (blk + 2 < nblocks) { VDB_DLOAD(a2, b2, blk + 2) }
        VDB_DMFMA(a0, b0)
        if (blk + 1 >= nblocks) break;
        if (blk + 3 < nblocks) { VDB_DLOAD(a0, b0, blk + 3) }
        VDB_DMFMA(a1, b1)
        if (blk + 2 >= nblocks) break;
        if (blk + 4 < nblocks) { VDB_DLOAD(a1, b1, blk + 4) }
        VDB_DMFMA(a2, b2)
    }
#undef VDB_DLOAD
#undef VDB_DMFMA
    // C/D layout: column (query) = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const uint32_t q = blockIdx.y * 32 + c;
    uint64_t* out = p.keys + (size_t)q * p.key_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t j = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (j < p.n_sample) {
            uint32_t row = sample_row(j, p.n_sample, p.n_rows);
            bool ok = p.rowmask ? ((p.rowmask[row >> 5] >> (row & 31)) & 1u) : true;
            float sc = fmaf(acc[r], p.alpha[row], p.beta[row]);
            out[j] = ok ? make_key(sc, row) : EMPTY_KEY;
        }
    }
}
void launch_dense_scores(const DenseParams& p, hipStream_t s) {
    uint32_t ntiles = (p.n_sample + 31) / 32;
    if (!ntiles) return;
    hipLaunchKernelGGL(dense_scores_kernel, dim3((ntiles + 3) / 4, p.nq_pad / 32), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------------------------
// Per-query radix select: the kk smallest of n 64-bit keys (all distinct: the low word is
// a row), written sorted ascending.  One 256-thread workgroup per query; 8 MSB-first
// digit passes find the kk-th key exactly, one pass collects, a bitonic network sorts.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SEL_LDS_KEYS = 16384;  // keys cached in LDS when n fits (128 KB)
constexpr uint32_t SEL_MAX_KK = 2048;
constexpr uint32_t SEL_THREADS = 1024;   // 16 waves per query: the passes are latency-bound

// EMIT mode: selected key i of query q as a final (id, distance) result
__device__ __forceinline__ void select_emit(const SelectParams& p, uint32_t q, uint32_t i, uint64_t key, bool valid) {
    const size_t o = (size_t)q * p.emit_stride + i;
    if (valid) {
        const uint32_t rk = (uint32_t)key;
        const uint32_t row = p.emit_rank2row ? p.emit_rank2row[rk] : rk;
        p.emit_ids[o] = p.emit_row_ids[row];
        p.emit_dists[o] = ordered_to_f32((uint32_t)(key >> 32));
    } else {
        p.emit_ids[o] = ~0ull;
        p.emit_dists[o] = __uint_as_float(0x7fc00000u);
    }
}

__global__ __launch_bounds__(SEL_THREADS) void select_kernel(SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sdyn[];
    uint64_t* sKeys = sdyn;                       // [SEL_LDS_KEYS]
    uint64_t* sOut = sdyn + SEL_LDS_KEYS;         // [SEL_MAX_KK]
    __shared__ uint32_t sHist[256];
    __shared__ uint32_t sDigit, sRemain, sBucket, sOutCnt, sValid, sN;
    __shared__ unsigned long long sMinMax[2];

    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint64_t* keys = p.keys + (size_t)q * p.stride;
    uint32_t n = p.n_fixed;
    bool cached;
    const bool has_lo = p.lo_excl != nullptr;
    const uint64_t lo = has_lo ? p.lo_excl[q] : 0ull;
    if (tid == 0) { sOutCnt = 0; sValid = 0; sN = 0; sMinMax[0] = ~0ull; sMinMax[1] = 0ull; }
    __syncthreads();
    uint32_t myvalid = 0;
    if (p.n_sub) {
        // gather the query's private sub-pools (written by the fused kernels) into LDS.  The pools are sparse
        // (a few keys in each of hundreds of sub-pools), so the gather is driven by the counts: one thread per
        // sub-pool reads its count (coalesced), a wave scan + one LDS atomic per wave allots the destination
        // range, and the thread copies its keys -- every load address is known after the count, so all of a
        // thread's loads are in flight together.
        const uint32_t* sc = p.sub_counts + (size_t)q * p.n_sub;
        const uint64_t* base = p.keys + (size_t)q * p.n_sub * p.capl;
        // (bf16 tier, several 256-query blocks in one launch: the block's pools, the query's place inside its block)
        const uint32_t ql = p.wg_major ? (q & 255u) : q;
        const uint32_t* wg_cnts = p.sub_counts + (size_t)(q >> 8) * p.blk_cnts;
        const uint64_t* wg_keys = p.keys + (size_t)(q >> 8) * p.blk_keys;
        bool over = false;
        for (uint32_t i0 = 0; i0 < p.n_sub; i0 += SEL_THREADS) {
            const uint32_t i = i0 + tid;
            // (bf16 tier: counts workgroup-major like the keys -- four consecutive threads read one workgroup's 16 bytes)
            uint32_t c = i < p.n_sub ? (p.wg_major ? wg_cnts[((size_t)(i >> 2) * 256u + ql) * 4u + (i & 3u)] : sc[i]) : 0u;
            if (c > p.capl) { c = p.capl; over = true; }
            uint32_t incl = c;
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
            uint32_t wbase = 0;
            if (lane == 63 && incl) wbase = atomicAdd(&sN, incl);
            wbase = __shfl(wbase, 63);
            uint32_t pos = wbase + incl - c;
            // (bf16 tier: keys workgroup-major, sub-pool i = wg*4 + r of query q at ((wg*256 + q)*4 + r)*capl)
            const uint64_t* src = p.wg_major ? wg_keys + (((size_t)(i >> 2) * 256u + ql) * 4u + (i & 3u)) * p.capl : base + (size_t)i * p.capl;
            for (uint32_t j0 = 0; j0 < c; j0 += 4) {
                uint64_t k[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k[u] = (j0 + u < c) ? raw_to_key(src[j0 + u]) : EMPTY_KEY;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j0 + u < c) {
                        if (pos + j0 + u < SEL_LDS_KEYS) { sKeys[pos + j0 + u] = k[u]; myvalid += (k[u] != EMPTY_KEY); }
                        else over = true;
                    }
                }
            }
        }
        if (over && p.ovf) { p.ovf[q] = 1u; if (p.summary) atomicOr(p.summary, 2u); }
        __syncthreads();
        n = sN < SEL_LDS_KEYS ? sN : SEL_LDS_KEYS;
        cached = true;
    } else {
        if (p.counts) {
            uint32_t c = p.counts[q];
            if (c > p.cap) { c = p.cap; if (p.ovf && tid == 0) { p.ovf[q] = 1u; if (p.summary) atomicOr(p.summary, 2u); } }
            n = c;
        }
        cached = n <= SEL_LDS_KEYS;
        for (uint32_t i = tid; i < n; i += SEL_THREADS) {
            uint64_t k = keys[i];
            if (has_lo && k <= lo) k = EMPTY_KEY;          // already emitted by an earlier chunk
            if (cached) sKeys[i] = k;
            myvalid += (k != EMPTY_KEY);
        }
    }
    for (int o = 32; o > 0; o >>= 1) myvalid += __shfl_xor(myvalid, o);
    if (lane == 0 && myvalid) atomicAdd(&sValid, myvalid);
    __syncthreads();
    const uint32_t nvalid = sValid;
    const uint32_t kk = nvalid < p.kk ? nvalid : p.kk;
    if (p.flag_truncation && nvalid > p.kk && p.ovf && tid == 0) { p.ovf[q] = 1u; if (p.summary) atomicOr(p.summary, 2u); }
    uint64_t* out = p.out_keys + (size_t)q * p.out_stride;
    if (p.emit_ids && tid == 0) { p.emit_counts[q] = kk; if (p.emit_status) p.emit_status[q] = *p.emit_status_in; }
    if (kk == 0) {
        for (uint32_t i = tid; i < p.kk; i += SEL_THREADS) out[i] = EMPTY_KEY;
        if (p.emit_ids) for (uint32_t i = tid; i < p.emit_stride; i += SEL_THREADS) select_emit(p, q, i, 0, false);
        if (tid == 0) { p.out_cnt[q] = 0; if (p.out_thr) p.out_thr[q] = __uint_as_float(0x7f800000u); }
        return;
    }
    // ---- short lists (the screening tier's pools: a few hundred keys): rank by counting.  Every thread owns one key and counts
    // the keys below it -- the keys of a query are distinct, so the counts are the sorted positions -- reading the list from LDS
    // at one address per step for the whole workgroup (a broadcast).  No digit passes, no collection, no 36-step bitonic
    // network with a barrier per step: gather-select of 244 keys per query 21 -> 12 us at config 2.
    if (cached && n <= SEL_THREADS) {
        const uint64_t mine = tid < n ? sKeys[tid] : EMPTY_KEY;
        uint32_t rank = 0;
        if (mine != EMPTY_KEY) {
            uint32_t j = 0;
            for (; j + 2 <= n; j += 2) {
                const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(sKeys + j);
                rank += (k2.x < mine) + (k2.y < mine);
            }
            if (j < n) rank += sKeys[j] < mine;
        }
        if (mine != EMPTY_KEY && rank < kk) out[rank] = mine;
        for (uint32_t i = kk + tid; i < p.kk; i += SEL_THREADS) out[i] = EMPTY_KEY;
        if (p.emit_ids) {
            if (mine != EMPTY_KEY && rank < kk) select_emit(p, q, rank, mine, true);
            for (uint32_t i = kk + tid; i < p.emit_stride; i += SEL_THREADS) select_emit(p, q, i, 0, false);
        }
        if (mine != EMPTY_KEY && rank == kk - 1) {                  // the kk-th smallest key: exactly one thread
            p.out_cnt[q] = kk;
            if (p.out_last) p.out_last[q] = mine;
            if (p.out_thr) {
                float t = (kk == p.kk) ? ordered_to_f32((uint32_t)(mine >> 32)) : __uint_as_float(0x7f800000u);
                if (p.shift_g) {
                    const uint32_t mb = *p.shift_m_bits;
                    if (mb) t = fmaf(-p.shift_g[q], __uint_as_float(~mb), t);
                }
                p.out_thr[q] = t;
            }
        }
        return;
    }
    // ---- find the kk-th smallest key
    // Scores of one query are concentrated (same sign and exponent, often the same leading mantissa bits), so the
    // leading bytes all keys share are found first (one min / max reduction over the LDS-resident keys) and the
    // digit passes start below them.
    int b_first = 7;
    uint64_t prefix = 0;
    if (cached) {
        uint64_t kmin = EMPTY_KEY, kmax = 0;
        for (uint32_t i = tid; i < n; i += SEL_THREADS) {
            uint64_t k = sKeys[i];
            if (k != EMPTY_KEY) { kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            uint64_t a = __shfl_xor(kmin, o), b2 = __shfl_xor(kmax, o);
            kmin = a < kmin ? a : kmin;
            kmax = b2 > kmax ? b2 : kmax;
        }
        if (lane == 0) { atomicMin(&sMinMax[0], kmin); atomicMax(&sMinMax[1], kmax); }
        __syncthreads();
        const uint64_t diff = sMinMax[0] ^ sMinMax[1];
        if (diff) {
            b_first = (63 - __clzll((long long)diff)) >> 3;        // highest byte in which two keys differ
            if (b_first < 7) prefix = sMinMax[0] >> (8 * (b_first + 1));
        } else {
            b_first = 0;                                           // a single distinct key
            prefix = sMinMax[0] >> 8;
        }
    }
    uint32_t remain = kk;
    uint32_t inbucket = nvalid;          // keys still matching the prefix
    for (int b = b_first; b >= 0; --b) {
        if (tid < 256) sHist[tid] = 0;
        __syncthreads();
        const int shift = 8 * b;
        // Scores are concentrated, so in the first passes nearly every key lands in one or two bins:
        // aggregate equal digits inside the wave (one LDS atomic per distinct digit) while many keys
        // still take part; plain atomics once the bucket is small.
        const bool aggregate = (b == b_first) && inbucket > 512;   // first digit examined: usually few distinct values
        for (uint32_t i = tid; i < n; i += SEL_THREADS) {
            uint64_t k = cached ? sKeys[i] : keys[i];
            if (!cached && has_lo && k <= lo) k = EMPTY_KEY;
            bool in = (k != EMPTY_KEY) && (b == 7 || (k >> (shift + 8)) == prefix);   // (all valid keys share a skipped prefix)
            uint32_t d = (uint32_t)(k >> shift) & 255u;
            if (aggregate) {
                unsigned long long active = __ballot(in);
                while (active) {
                    int leader = __ffsll((long long)active) - 1;
                    uint32_t dl = __shfl(d, leader);
                    unsigned long long same = __ballot(in && d == dl);
                    if ((int)lane == leader) atomicAdd(&sHist[dl], (uint32_t)__popcll(same));
                    active &= ~same;
                }
            } else if (in) {
                atomicAdd(&sHist[d], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            uint32_t c0 = sHist[4 * lane], c1 = sHist[4 * lane + 1], c2 = sHist[4 * lane + 2], c3 = sHist[4 * lane + 3];
            uint32_t sum = c0 + c1 + c2 + c3, incl = sum;
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
            unsigned long long hit = __ballot(incl >= remain);
            int first = __ffsll((long long)hit) - 1;
            if ((int)lane == first) {
                uint32_t r = remain - (incl - sum);
                uint32_t d = 0, cd = c0;
                if (r > c0) { r -= c0; d = 1; cd = c1; if (r > c1) { r -= c1; d = 2; cd = c2; if (r > c2) { r -= c2; d = 3; cd = c3; } } }
                sDigit = 4 * lane + d;
                sRemain = r;
                sBucket = cd;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | sDigit;
        remain = sRemain;
        inbucket = sBucket;
        __syncthreads();
    }
    const uint64_t pivot = prefix;
    // ---- collect keys <= pivot (exactly kk of them, keys are distinct)
    for (uint32_t i = tid; i < n; i += SEL_THREADS) {
        uint64_t k = cached ? sKeys[i] : keys[i];
        if (!cached && has_lo && k <= lo) k = EMPTY_KEY;
        if (k <= pivot) {     // EMPTY_KEY is the maximum and pivot is a real key, so it never passes
            uint32_t slot = atomicAdd(&sOutCnt, 1u);
            if (slot < SEL_MAX_KK) sOut[slot] = k;
        }
    }
    __syncthreads();
    uint32_t P = 2;
    while (P < kk) P <<= 1;
    for (uint32_t i = kk + tid; i < P; i += SEL_THREADS) sOut[i] = EMPTY_KEY;
    __syncthreads();
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < P / 2; t += SEL_THREADS) {
                uint32_t lo = 2 * t - (t & (stride - 1));      // index with bit `stride` cleared
                uint32_t hi2 = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t a = sOut[lo], b2 = sOut[hi2];
                if ((a > b2) == up) { sOut[lo] = b2; sOut[hi2] = a; }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < p.kk; i += SEL_THREADS) out[i] = i < kk ? sOut[i] : EMPTY_KEY;
    if (p.emit_ids) for (uint32_t i = tid; i < p.emit_stride; i += SEL_THREADS) select_emit(p, q, i, i < kk ? sOut[i] : 0, i < kk);
    if (tid == 0) {
        p.out_cnt[q] = kk;
        if (p.out_last) p.out_last[q] = sOut[kk - 1];
        if (p.out_thr) {
            float t = (kk == p.kk) ? ordered_to_f32((uint32_t)(sOut[kk - 1] >> 32)) : __uint_as_float(0x7f800000u);
            if (p.shift_g) {
                const uint32_t mb = *p.shift_m_bits;
                if (mb) t = fmaf(-p.shift_g[q], __uint_as_float(~mb), t);
            }
            p.out_thr[q] = t;
        }
    }
}
constexpr uint32_t TS_THREADS = 256, TS_PER = 16;                 // up to 4096 keys per query
__global__ __launch_bounds__(TS_THREADS) void thr_select_kernel(SelectParams p) {
    __shared__ uint32_t sHist[256];
    __shared__ uint32_t sDigit, sRemain, sValid;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint64_t* keys = p.keys + (size_t)q * p.stride;
    const uint32_t n = p.n_fixed;
    // the ORDER of two keys with the same score word does not matter for the VALUE of the kk-th smallest score, so only the
    // high words take part (an empty slot, ~0, sorts last)
    uint32_t v[TS_PER];
    uint32_t myvalid = 0;
#pragma unroll
    for (uint32_t u = 0; u < TS_PER; ++u) {
        const uint32_t i = u * TS_THREADS + tid;
        uint32_t hi = 0xffffffffu;
        if (i < n) { const uint64_t key = keys[i]; if (key != EMPTY_KEY) { hi = (uint32_t)(key >> 32); ++myvalid; } }
        v[u] = hi;
    }
    if (tid == 0) sValid = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) myvalid += __shfl_xor(myvalid, o);
    if (lane == 0 && myvalid) atomicAdd(&sValid, myvalid);
    __syncthreads();
    const uint32_t nvalid = sValid;
    float t = __uint_as_float(0x7f800000u);                        // fewer than kk valid keys: +inf, everything passes
    if (nvalid >= p.kk && p.kk > 0) {
        uint32_t prefix = 0, remain = p.kk;
        for (int b = 3; b >= 0; --b) {
            sHist[tid] = 0;
            __syncthreads();
            const int shift = 8 * b;
#pragma unroll
            for (uint32_t u = 0; u < TS_PER; ++u) {
                const bool in = v[u] != 0xffffffffu && (b == 3 || (v[u] >> (shift + 8)) == prefix);
                if (in) atomicAdd(&sHist[(v[u] >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                const uint32_t c0 = sHist[4 * lane], c1 = sHist[4 * lane + 1], c2 = sHist[4 * lane + 2], c3 = sHist[4 * lane + 3];
                const uint32_t sum = c0 + c1 + c2 + c3;
                uint32_t incl = sum;
                for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if ((int)lane >= o) incl += x; }
                const unsigned long long hit = __ballot(incl >= remain);
                const int first = __ffsll((long long)hit) - 1;
                if ((int)lane == first) {
                    uint32_t r = remain - (incl - sum), d = 0;
                    if (r > c0) { r -= c0; d = 1; if (r > c1) { r -= c1; d = 2; if (r > c2) { r -= c2; d = 3; } } }
                    sDigit = 4 * lane + d;
                    sRemain = r;
                }
            }
            __syncthreads();
            prefix = (prefix << 8) | sDigit;
            remain = sRemain;
            __syncthreads();
        }
        t = ordered_to_f32(prefix);
    }
    if (tid == 0) {
        if (p.shift_g) {
            const uint32_t mb = *p.shift_m_bits;
            if (mb) t = fmaf(-p.shift_g[q], __uint_as_float(~mb), t);
        }
        p.out_thr[q] = t;
    }
}
void launch_thr_select(const SelectParams& p, uint32_t nq, hipStream_t s) {
    if (!nq) return;
    if (p.n_fixed > TS_THREADS * TS_PER || p.n_sub || p.counts || p.lo_excl || !p.out_thr) { launch_select(p, nq, s); return; }
    hipLaunchKernelGGL(thr_select_kernel, dim3(nq), dim3(TS_THREADS), 0, s, p);
}

void launch_select(const SelectParams& p, uint32_t nq, hipStream_t s) {
    if (!nq) return;
    hipLaunchKernelGGL(select_kernel, dim3(nq), dim3(SEL_THREADS), (SEL_LDS_KEYS + SEL_MAX_KK) * sizeof(uint64_t), s, p);
}

// ---------------------------------------------------------------------------------------------
// Exact re-rank of the candidates of each query + certification + final ordering.
// One 512-thread workgroup per query.  Candidates arrive sorted by ranking score; their distances are
// computed in the reference's exact operation order (bit-identical to the CPU oracle), one candidate per
// thread, rows staged through LDS in chunks.
// ADAPTIVE DEPTH: the first kp_first candidates are re-ranked, then the result is tested: every row not
// yet re-ranked has ranking score >= T (the score of the first candidate not re-ranked; the filter
// threshold once the whole pool is used up).  From T a lower bound LB on such a row's exact distance
// follows with the tier's error bound; if the k-th exact distance so far is < LB no other row
// can enter the top k and the result is the oracle's.  Otherwise kp_step more candidates are re-ranked and
// the test repeated, up to the kp candidates the select delivered; then cert[q] = 0 and the host hands
// the query to the next tier.
// ---------------------------------------------------------------------------------------------
// The certification test: every row not re-ranked has ranking score >= T; is the k-th exact distance ek below the
// lower bound that T implies for such a row's exact distance?  (DESIGN.md "certified top-k")
// UNDERFLOW (found by tests/test_gpu_certificate.py "subnormal"): every bound below takes f32 norms and sums as accurate
// to a few K 2^-24 RELATIVE.  That fails when the squares underflow f32 (|x| below ~1e-19: fold(x*x) loses its low terms
// or all of them), so
//   * a query whose exact-order norm is below 2^-40 is never certified by an MFMA tier (it ends in the exact scan);
//   * under Cosine an index holding a live-or-dead row with 0 < |d| < 2^-40 certifies nothing (same consequence);
//   * every test gives away an ABSOLUTE floor of K 2^-140 in product units for the f32 operations that underflowed
//     along the way (each loses less than 2^-149; the MFMA accumulators themselves keep denormals -- measured).
constexpr double CERT_TINY_NORM = 9.094947017729282e-13;        // 2^-40
__device__ __forceinline__ double cert_floor(const RerankParams& p) { return (double)p.ld * 7.174648137343064e-43; }   // ld 2^-140
// The test has the form   lhs(e_k) < a + b * (T - |T| tslack)   with lhs = e_k, or e_k^2 (1 + eps) under Euclid.  a, b and
// tslack depend on the QUERY only (norms, rounding-error norm, per-index scalars): they are computed once per workgroup --
// double-precision square roots and divisions -- and every candidate thread then evaluates the test in three flops.
struct CertConsts { double a, b, tslack, lhs_scale; int square; int ok; };
__device__ __forceinline__ CertConsts cert_consts(const RerankParams& p, uint32_t q, double qn) {
    CertConsts c{0.0, 1.0, 0.0, 1.0, 0, 1};
    const double eps = (double)p.eps_coef;
    const double ndmax = sqrt((double)__uint_as_float(p.nd2max_bits[0]));
    const double fl = cert_floor(p);
    if (!(qn >= CERT_TINY_NORM)) { c.ok = 0; return c; }
    if (p.metric == COSINE) {
        const uint32_t mb = p.nd2max_bits[5];
        if (mb && !(__uint_as_float(~mb) >= (float)(CERT_TINY_NORM * CERT_TINY_NORM))) { c.ok = 0; return c; }
    }
    if (p.metric == EUCLID) { c.square = 1; c.lhs_scale = 1.0 + eps; }
    if (p.qerr && p.lb_scores) {
        // bf16 screening tier, Dot / Euclid: T is a LOWER-BOUND score (FusedBf16Params::margin) -- the bf16 rounding of
        // row and query, the MFMA accumulation and the row's share of the f32-fold budget were subtracted per row in the
        // kernel, so only the query's own terms are left here and no per-index maximum enters: one huge-norm row
        // loosens nobody's certificate but its own.  tslack: the f32 rounding of the margin fma.
        //   Dot:     e_k < Tl - fl                                   Euclid:  e_k^2 (1 + eps) < Tl + |q|^2 (1 - eps) - 4 fl
        c.tslack = 2.4e-7;
        c.a = p.metric == DOT ? -fl : qn * qn * (1.0 - eps) - 4.0 * fl;
        return c;
    }
    double E = 0.0, Ec = 0.0;
    if (p.qerr) {
        // bf16 screening tier (Cosine: the relative rounding error of a row is bounded by 2^-9 whatever its norm, so the
        // per-index maximum of it is a local quantity already; Dot / Euclid land here only for the raw-score diagnostics).
        // With e_q = q - bf16(q) (known) and e_d = d - bf16(d):
        //   dot(q,d) - dot(bf16 q, bf16 d) = e_q.d + bf16(q).e_d ,  |.| <= |e_q||d| + |bf16 q||e_d|   (Cauchy-Schwarz)
        // plus the f32 accumulation inside the MFMAs (c_acc |q||d|).  |bf16 q| <= 1.004 |q|; 1 % covers the f32
        // evaluation of the norms.  eps is the f32 tier's coefficient (oracle fold + fma chain).
        const double eq = (double)p.qerr[q];
        const double emax = sqrt((double)__uint_as_float(p.nd2max_bits[2]));
        const double rmax = sqrt((double)__uint_as_float(p.nd2max_bits[3]));
        const double cacc = (double)p.c_acc;
        E = 1.01 * (eq * ndmax + 1.004 * qn * emax) + cacc * qn * ndmax;
        Ec = 1.01 * (eq / qn + 1.004 * rmax) + cacc;
    }
    //   Dot:     e_k < T - E - eps |q| max|d| - fl
    //   Cosine:  e_k < 1 + T / |q| - Ec - eps - fl / (|q| 2^-40)
    //   Euclid:  e_k^2 (1 + eps) < T + |q|^2 - 2 E - eps (|q| + max|d|)^2 - 4 fl
    if (p.metric == DOT) c.a = -E - eps * qn * ndmax - fl;
    else if (p.metric == COSINE) { c.a = 1.0 - Ec - eps - fl / (qn * CERT_TINY_NORM); c.b = 1.0 / qn; }
    else { const double s = qn + ndmax; c.a = qn * qn - 2.0 * E - eps * s * s - 4.0 * fl; }
    return c;
}
__device__ __forceinline__ bool cert_eval(const CertConsts& c, float T, double ek) {
    if (!c.ok) return false;
    const double Tl = (double)T - fabs((double)T) * c.tslack;
    const double lhs = c.square ? ek * ek * c.lhs_scale : ek;
    return lhs < c.a + c.b * Tl;
}
__device__ __forceinline__ bool cert_test(const RerankParams& p, uint32_t q, float T, double ek, double qn) {
    return cert_eval(cert_consts(p, q, qn), T, ek);
}

// The inverse of cert_test: the smallest score T* such that cert_test(T) holds for every T > T*.  Every row whose
// ranking score exceeds it is PROVEN to lie beyond the k-th exact distance ek, so a filter pass with T* (rounded up) as
// its threshold keeps every row that can still matter (the "re-threshold" pass of vdb_flat.cpp).
__device__ __forceinline__ float score_cut(const RerankParams& p, uint32_t q, double ek, double qn) {
    const double eps = (double)p.eps_coef;
    const double ndmax = sqrt((double)__uint_as_float(p.nd2max_bits[0]));
    double E = 0.0, Ec = 0.0;
    if (p.qerr) {
        const double eq = (double)p.qerr[q];
        const double emax = sqrt((double)__uint_as_float(p.nd2max_bits[2]));
        const double rmax = sqrt((double)__uint_as_float(p.nd2max_bits[3]));
        const double cacc = (double)p.c_acc;
        E = 1.01 * (eq * ndmax + 1.004 * qn * emax) + cacc * qn * ndmax;
        Ec = 1.01 * (eq / qn + 1.004 * rmax) + cacc;
    }
    const double fl = cert_floor(p);
    if (!(qn >= CERT_TINY_NORM)) return __uint_as_float(0x7fc00000u);          // not certifiable on an MFMA tier: no cut
    if (p.metric == COSINE) {
        const uint32_t mb = p.nd2max_bits[5];
        if (mb && !(__uint_as_float(~mb) >= (float)(CERT_TINY_NORM * CERT_TINY_NORM))) return __uint_as_float(0x7fc00000u);
    }
    double t;
    if (p.qerr && p.lb_scores) {
        t = p.metric == DOT ? ek + fl : ek * ek - qn * qn + eps * (qn * qn + ek * ek) + 4.0 * fl;
        t += fabs(t) * 2.4e-7;
    }
    else if (p.metric == DOT) t = ek + E + eps * qn * ndmax + fl;
    else if (p.metric == COSINE) t = (ek - 1.0 + Ec + eps + fl / (qn * CERT_TINY_NORM)) * qn;
    else { const double s = qn + ndmax; t = ek * ek - qn * qn + 2.0 * E + eps * (s * s + ek * ek) + 4.0 * fl; }
    t += fabs(t) * 1e-6 + 1e-30;                                // slack: looser is safe
    float f = (float)t;
    if ((double)f < t) f = __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1u : (uint32_t)-1));   // round towards +inf
    return f;
}

// ---- certificate diagnostics (tests/test_gpu_certificate.py through vdb_flat_debug_*)
__global__ __launch_bounds__(256) void pool_to_dense_kernel(const uint64_t* pool, const uint32_t* pool_cnt, uint32_t n_sub,
                                                            uint32_t capl, uint32_t n_rows, float* dense) {
    const uint32_t q = blockIdx.y, i = blockIdx.x;                 // sub-pool i = wg*4 + r of query q
    uint32_t c = pool_cnt[((size_t)(i >> 2) * 256u + q) * 4u + (i & 3u)];
    if (c > capl) c = capl;
    const uint64_t* src = pool + (((size_t)(i >> 2) * 256u + q) * 4u + (i & 3u)) * capl;
    for (uint32_t j = threadIdx.x; j < c; j += blockDim.x) {
        const uint64_t raw = src[j];
        const uint32_t row = (uint32_t)raw;
        if (row < n_rows) dense[(size_t)q * n_rows + row] = __uint_as_float((uint32_t)(raw >> 32));
    }
}
void launch_pool_to_dense(const uint64_t* pool, const uint32_t* pool_cnt, uint32_t n_sub, uint32_t capl, uint32_t nq,
                          uint32_t n_rows, float* dense, hipStream_t s) {
    if (!nq || !n_sub) return;
    hipLaunchKernelGGL(pool_to_dense_kernel, dim3(n_sub, nq), dim3(256), 0, s, pool, pool_cnt, n_sub, capl, n_rows, dense);
}
__global__ __launch_bounds__(256) void cert_probe_kernel(RerankParams p, const uint32_t* qi, const float* T, const float* ek,
                                                         uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = qi[i];
    out[i] = cert_test(p, q, T[i], (double)ek[i], (double)p.qnorm[q]) ? 1u : 0u;
}
void launch_cert_probe(const RerankParams& p, const uint32_t* qi, const float* T, const float* ek, uint32_t n, uint32_t* out,
                       hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(cert_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, qi, T, ek, n, out);
}

typedef __attribute__((address_space(3))) void* rr_lds_t;
typedef const __attribute__((address_space(1))) void* rr_glb_t;
constexpr uint32_t RR_MAX = 512;        // candidates per query at most (= RR_THREADS: one thread per candidate)
constexpr uint32_t RR_THREADS = 512;

__global__ __launch_bounds__(RR_THREADS) void rerank_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) float sRows[];   // query row + `chunk` candidate rows
    __shared__ uint32_t sDist[RR_MAX];    // ordered exact distance
    __shared__ uint64_t sId[RR_MAX];
    __shared__ uint32_t sRowIdx[RR_MAX];
    __shared__ uint32_t sAnyNan, sNanKey, sNext, sRealW[RR_THREADS / 64];
    __shared__ CertConsts sCert;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef VDB_DIAG
    // phase stamps for tools/rerank_depth.sh (diagnostics build, VDB_RR_DEPTH): p.depth[gridDim.x .. ] as 8 x u64 per query
#define VDB_STAMP(PH) if (p.depth && tid == 0) reinterpret_cast<uint64_t*>(p.depth + 256)[(size_t)q * 8 + (PH)] = __builtin_amdgcn_s_memrealtime();
#else
#define VDB_STAMP(PH)
#endif
    VDB_STAMP(0)
    const uint32_t cnt = p.cand_cnt[q] < p.kp ? p.cand_cnt[q] : p.kp;
    const uint64_t* cand = p.cand + (size_t)q * p.cand_stride;
    if (tid == 0) { sAnyNan = 0; sNanKey = 0; sNext = 0; }
    if (tid == RR_THREADS - 1) sCert = cert_consts(p, q, (double)p.qnorm[q]);     // once per query, beside the row staging (published by the barriers below)
    if (tid < RR_MAX) {
        uint32_t row = 0xffffffffu;
        if (tid < cnt) {
            uint64_t key = cand[tid];
            row = (uint32_t)key;
            if ((uint32_t)(key >> 32) == 0u) sNanKey = 1u;      // approximate score was NaN (benign race)
            bool ok = row < p.n_rows && (p.rowmask ? ((p.rowmask[row >> 5] >> (row & 31)) & 1u) : true);
            if (!ok) row = 0xffffffffu;
        }
        sRowIdx[tid] = row;
        sDist[tid] = 0xffffffffu;
        sId[tid] = ~0ull;
    }
    // the query row in LDS, then the slice area
    const uint32_t dimp = (p.dim + 3) & ~3u;
    const uint32_t area = p.lds_chunk;                            // floats available for candidate slices
    float* sQ = sRows;
    float* sR = sRows + p.lds_row_stride;
    const float* gq = p.qp + (size_t)q * p.ld;                    // zero padded up to ld >= dimp
    for (uint32_t i = tid * 4; i < dimp; i += RR_THREADS * 4) *reinterpret_cast<float4*>(sQ + i) = *reinterpret_cast<const float4*>(gq + i);
    __syncthreads();
    const uint32_t nwaves = RR_THREADS / 64;
    const float qn_f = p.qnorm[q];
    VDB_STAMP(1)

    uint32_t processed = 0;
    uint32_t target = cnt < p.kp_first ? cnt : p.kp_first;
    // PREDICTED depth of the first round (plain scores only: Cosine on the screening tier, and the f32 tier).  The candidates
    // arrive sorted by ranking score, and the score of candidate k + 1 gives an estimate of the k-th exact distance before any
    // row is fetched (Cosine d = 1 + s / |q|, Euclid d^2 = s + |q|^2, Dot d = s); the first candidate m that the production
    // test would certify against THAT distance is where round 1 should end.  The estimate only chooses a depth: the round
    // is tested with the exact k-th distance as before, and a query whose estimate was too optimistic takes a second round
    // as before.  Easy queries then fetch 24-40 rows instead of a fixed 48, hard ones finish in one round of 70-100.
    if (!p.lb_scores && cnt > p.k + 8 && p.k > 0) {
        if (tid == 0) sNext = 0xffffffffu;
        __syncthreads();
        const uint32_t ke = p.k < cnt ? p.k : cnt - 1;                // the (k+1)-th best ranking score: a slightly pessimistic k-th distance
        const uint64_t kkey = cand[ke];
        const float sk = ordered_to_f32((uint32_t)(kkey >> 32));
        const double qnd = (double)qn_f;
        double ek_est = p.metric == DOT ? (double)sk : p.metric == COSINE ? 1.0 + (double)sk / qnd : sqrt(fmax(0.0, (double)sk + qnd * qnd));
        if ((uint32_t)(kkey >> 32) != 0u && ek_est == ek_est && tid >= p.k && tid < cnt) {
            const float T = ordered_to_f32((uint32_t)(cand[tid] >> 32));
            if (cert_eval(sCert, T, ek_est)) atomicMin(&sNext, tid);
        }
        __syncthreads();
        const uint32_t m = sNext;
        if (m != 0xffffffffu) {
            uint32_t t = (m + 4u + 7u) & ~7u;                       // a little deeper than predicted, in whole 8s
            const uint32_t lo = (p.k + 6u + 7u) & ~7u;
            if (t < lo) t = lo;
            if (t > 160u) t = 160u;
            target = t < cnt ? t : cnt;
        }
        __syncthreads();
    }
    uint32_t nout = 0;
    uint32_t cert = 1;
    bool cut_ok = false;                                          // a k-th exact distance exists and no NaN / ineligible candidate was seen
    double cut_ek = 0.0;
    while (true) {
        // ---- exact distances of candidates [processed, target): ALL of them at once, one thread per candidate, the rows
        // staged through LDS in K SLICES as wide as the slice area allows for that many rows (48 rows of 768: one slice = the
        // whole row; 96 rows: two slices of 384; 144 rows: three of 256).  A thread carries its partial sum from slice to
        // slice in a register -- the fold is the reference's sequential one, element by element, whatever the slicing -- so
        // the depth of a round is no longer capped by what fits into LDS as whole rows, and a round costs one staging
        // latency per slice instead of one per 48 candidates.
        {
            const uint32_t n = target - processed;                  // <= RR_MAX
            uint32_t W = n ? (area / n - 4u) & ~15u : 16u;          // slice width: a multiple of the fold's unroll, + the bank padding below
            if (W > dimp) W = (dimp + 15u) & ~15u;
            if (W < 16) W = 16;                                     // (the launcher sizes the area for 512 rows of 16 + 4)
            const uint32_t Wp = W + ((W % 8 == 0) ? 4 : 0);         // row stride: 16 lanes' b128 reads tile the banks
            const uint32_t myrow = tid < n ? sRowIdx[processed + tid] : 0xffffffffu;
            float acc = 0.0f;
            for (uint32_t k0 = 0; n && k0 < dimp; k0 += W) {
                const uint32_t wlen = dimp - k0 < W ? dimp - k0 : W;           // floats of this slice (a multiple of 4)
                const uint32_t vpr = wlen / 4, bpr = (vpr + 63) / 64;
                // stage the slice by LDS-DMA (global_load_lds_dwordx4: 1 KB of a row per wave instruction, no VGPR round
                // trip), every piece in flight at once; the barrier's vmcnt(0) waits for them
                for (uint32_t u = wv; u < n * bpr; u += nwaves) {
                    const uint32_t r = u / bpr, b = u % bpr, c4 = b * 64 + lane;
                    const uint32_t row = sRowIdx[processed + r];
                    if (row != 0xffffffffu && c4 < vpr)
                        __builtin_amdgcn_global_load_lds((rr_glb_t)(p.rows + (size_t)row * p.ld + k0 + 4 * c4),
                                                         (rr_lds_t)(sR + (size_t)r * Wp + 256 * b), 16, 0, 0);
                }
                __syncthreads();
                if (myrow != 0xffffffffu) {
                    const uint32_t flen = (k0 + wlen > p.dim) ? p.dim - k0 : wlen;   // the fold stops at dim, not at the padding
                    acc = p.metric == EUCLID ? fold_sqdiff_part(sQ + k0, sR + (size_t)tid * Wp, flen, acc)
                                             : fold_dot_part(sQ + k0, sR + (size_t)tid * Wp, flen, acc);
                }
                __syncthreads();
            }
            if (myrow != 0xffffffffu) {
                const float dist = distance_from_fold(p.metric, acc, qn_f, p.nd[myrow]);
                if (dist != dist) sAnyNan = 1u;
                sDist[processed + tid] = f32_to_ordered(dist);
                sId[processed + tid] = p.row_ids[myrow];
            }
            __syncthreads();
        }
        if (processed == 0) { VDB_STAMP(2) }
        VDB_STAMP(5)
        processed = target;
        // (ranking by counting -- what the gather-select now does for its short lists -- was tried here too: 3.8 us per round
        // against 2.8 for this network at 48 pairs; the 96-bit (distance, id) compare costs more than the barrier-free steps save)
        // ---- bitonic sort of the first P >= processed (dist, id) pairs, ascending; unused slots hold the maximum
        uint32_t P = 32;
        while (P < processed) P <<= 1;
#define VDB_CEX()                                                                                      \
        {                                                                                              \
            uint32_t lo = 2 * tid - (tid & (stride - 1));                                              \
            uint32_t hi = lo + stride;                                                                 \
            bool up = ((lo & size) == 0);                                                              \
            uint32_t da = sDist[lo], db = sDist[hi];                                                   \
            uint64_t ia = sId[lo], ib = sId[hi];                                                       \
            bool gt = da > db || (da == db && ia > ib);                                                \
            if (gt == up) { sDist[lo] = db; sDist[hi] = da; sId[lo] = ib; sId[hi] = ia; }              \
        }
        if (P <= 128) {
            // at most 64 compare-exchange pairs: ONE wave does the whole network.  LDS operations of a wave execute in
            // order, so the steps need no workgroup barrier between them -- only the compiler must keep their order.
            if (wv == 0)
                for (uint32_t size = 2; size <= P; size <<= 1)
                    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                        if (tid < P / 2) VDB_CEX()
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
            __syncthreads();
        } else {
            for (uint32_t size = 2; size <= P; size <<= 1)
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    if (tid < P / 2) VDB_CEX()
                    __syncthreads();
                }
        }
#undef VDB_CEX
        if (processed <= p.kp_first) { VDB_STAMP(3) }
        // number of real candidates so far (ineligible ones sorted to the end with id ~0)
        if (tid < RR_MAX) {
            unsigned long long b0 = __ballot(tid < processed && sId[tid] != ~0ull);
            if (lane == 0) sRealW[wv] = (uint32_t)__popcll(b0);
        }
        __syncthreads();
        uint32_t real = 0;
#pragma unroll
        for (uint32_t w8 = 0; w8 < RR_THREADS / 64; ++w8) real += sRealW[w8];
        nout = real < p.k ? real : p.k;
        // ---- certification / how much deeper to go
        const bool clean = !sNanKey && real == processed;          // no NaN score, no ineligible candidate so far
        const bool can_test = clean && nout == p.k && nout > 0;
        const double ek = can_test ? (double)ordered_to_f32(sDist[nout - 1]) : 0.0;
        cut_ok = can_test; cut_ek = ek;
        if (processed < cnt) {
            // every candidate not yet re-ranked tests "would the result be certified if the re-rank stopped just
            // before me": the first one that says yes is where the next round ends (ek can only shrink meanwhile)
            if (tid == 0) sNext = 0xffffffffu;
            __syncthreads();
            if (can_test && tid >= processed && tid < cnt) {
                const float T = ordered_to_f32((uint32_t)(cand[tid] >> 32));
                if (cert_eval(sCert, T, ek)) atomicMin(&sNext, tid);
            }
            __syncthreads();
            const uint32_t m = sNext;
            if (processed <= p.kp_first) { VDB_STAMP(4) }
            if (can_test && m == processed) { cert = 1; break; }   // certified at this depth
            if (!clean) { cert = 0; break; }                        // depth cannot repair a NaN score or an ineligible candidate
            if (can_test && m != 0xffffffffu) target = m;           // re-rank exactly up to the first certifying candidate
            else if (can_test) target = cnt;                        // none certifies: take the whole list, test against what lies beyond
            else target = processed + p.kp_step < cnt ? processed + p.kp_step : cnt;   // fewer than k real rows so far
            __syncthreads();
            continue;
        }
        // the whole candidate list is re-ranked
        {
            bool have_T = false;
            float T = 0.f;
            if (cnt == p.kp) { T = ordered_to_f32((uint32_t)(cand[p.kp - 1] >> 32)); have_T = true; }
            else if (p.thr && p.thr[q] < __uint_as_float(0x7f800000u)) {
                // the whole pool is re-ranked: every other row was rejected by the (finite) filter threshold
                T = p.thr[q]; have_T = true;
            }
            cert = 1;
            if (have_T && !(can_test && cert_eval(sCert, T, ek))) cert = 0;
        }
        break;
    }
    if (tid < p.k) {
        size_t o = (size_t)q * p.out_stride + tid;
        if (tid < nout) { p.out_ids[o] = sId[tid]; p.out_dists[o] = ordered_to_f32(sDist[tid]); }
        else { p.out_ids[o] = ~0ull; p.out_dists[o] = __uint_as_float(0x7fc00000u); }
    }
    if (tid == 0) {
        p.out_counts[q] = nout;
        if (sAnyNan) atomicOr(p.status, ST_NAN);
        p.cert[q] = cert;
        if (!cert) atomicOr(p.status + 1, 1u);             // summary word of the status block: some query needs the next tier
        if (p.thr_next) {
            float cut = __uint_as_float(0x7fc00000u);             // NaN: no cut known
            if (!cert && cut_ok) cut = score_cut(p, q, cut_ek, (double)qn_f);
            p.thr_next[q] = cut;
        }
        if (p.depth) p.depth[q] = processed;
    }
    VDB_STAMP(6)
#undef VDB_STAMP
}
// ---------------------------------------------------------------------------------------------
// Exhaustive re-rank (the re-threshold pass): EVERY key of the query's list is re-ranked in the reference's
// arithmetic, chunk by chunk; slots [0, k) of the sort area always hold the best k seen so far, each chunk is sorted
// in behind them and the area cut back to k.  The list is complete by construction (every row whose score is at or
// below the cut passed the filter), so the result is exact unless the list was truncated upstream.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RR_THREADS) void rerank_all_kernel(RerankParams p) {
    extern __shared__ __attribute__((aligned(16))) float sRows[];
    constexpr uint32_t AREA = 256;                                // k <= 112 plus a chunk of <= 64
    __shared__ uint32_t sDist[AREA];
    __shared__ uint64_t sId[AREA];
    __shared__ uint32_t sRowIdx[64];
    __shared__ uint32_t sAnyNan, sNanKey;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t cnt = p.cand_cnt[q] < p.cand_stride ? p.cand_cnt[q] : p.cand_stride;
    const uint64_t* cand = p.cand + (size_t)q * p.cand_stride;
    if (tid == 0) { sAnyNan = 0; sNanKey = 0; }
    if (tid < AREA) { sDist[tid] = 0xffffffffu; sId[tid] = ~0ull; }
    const uint32_t dimp = (p.dim + 3) & ~3u;
    const uint32_t ldp = p.lds_row_stride;
    uint32_t chunk = p.lds_chunk < 64 ? p.lds_chunk : 64;
    if (chunk > AREA - p.k) chunk = AREA - p.k;
    float* sQ = sRows;
    float* sR = sRows + ldp;
    const float* gq = p.qp + (size_t)q * p.ld;
    for (uint32_t i = tid * 4; i < dimp; i += RR_THREADS * 4) *reinterpret_cast<float4*>(sQ + i) = *reinterpret_cast<const float4*>(gq + i);
    __syncthreads();
    const uint32_t vpr = dimp / 4, bpr = (vpr + 63) / 64, nwaves = RR_THREADS / 64;
    const float qn_f = p.qnorm[q];
    for (uint32_t c0 = 0; c0 < cnt; c0 += chunk) {
        const uint32_t nthis = (cnt - c0 < chunk) ? cnt - c0 : chunk;
        if (tid < nthis) {
            const uint64_t key = cand[c0 + tid];
            uint32_t row = (uint32_t)key;
            if ((uint32_t)(key >> 32) == 0u) sNanKey = 1u;
            const bool ok = row < p.n_rows && (p.rowmask ? ((p.rowmask[row >> 5] >> (row & 31)) & 1u) : true);
            sRowIdx[tid] = ok ? row : 0xffffffffu;
        }
        __syncthreads();
        for (uint32_t u = wv; u < nthis * bpr; u += nwaves) {
            const uint32_t r = u / bpr, b = u % bpr, c4 = b * 64 + lane;
            const uint32_t row = sRowIdx[r];
            if (row != 0xffffffffu && c4 < vpr)
                __builtin_amdgcn_global_load_lds((rr_glb_t)(p.rows + (size_t)row * p.ld + 4 * c4),
                                                 (rr_lds_t)(sR + (size_t)r * ldp + 256 * b), 16, 0, 0);
        }
        __syncthreads();
        if (tid < nthis) {
            const uint32_t row = sRowIdx[tid];
            uint32_t od = 0xffffffffu;
            uint64_t id = ~0ull;
            if (row != 0xffffffffu) {
                const float dist = exact_distance(p.metric, sQ, sR + (size_t)tid * ldp, p.dim, qn_f, p.nd[row]);
                if (dist != dist) sAnyNan = 1u;
                od = f32_to_ordered(dist);
                id = p.row_ids[row];
            }
            sDist[p.k + tid] = od;
            sId[p.k + tid] = id;
        }
        __syncthreads();
        for (uint32_t size = 2; size <= AREA; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                if (tid < AREA / 2) {
                    uint32_t lo = 2 * tid - (tid & (stride - 1));
                    uint32_t hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint32_t da = sDist[lo], db = sDist[hi];
                    uint64_t ia = sId[lo], ib = sId[hi];
                    bool gt = da > db || (da == db && ia > ib);
                    if (gt == up) { sDist[lo] = db; sDist[hi] = da; sId[lo] = ib; sId[hi] = ia; }
                }
                __syncthreads();
            }
        if (tid >= p.k && tid < AREA) { sDist[tid] = 0xffffffffu; sId[tid] = ~0ull; }      // keep the best k only
        __syncthreads();
    }
    uint32_t nout = 0;
    {
        __shared__ uint32_t sCnt[RR_THREADS / 64];
        const unsigned long long b0 = __ballot(tid < p.k && sId[tid < AREA ? tid : 0] != ~0ull);
        if (lane == 0) sCnt[wv] = (uint32_t)__popcll(b0);
        __syncthreads();
        for (uint32_t w8 = 0; w8 < RR_THREADS / 64; ++w8) nout += sCnt[w8];
    }
    if (tid < p.k) {
        size_t o = (size_t)q * p.out_stride + tid;
        if (tid < nout) { p.out_ids[o] = sId[tid]; p.out_dists[o] = ordered_to_f32(sDist[tid]); }
        else { p.out_ids[o] = ~0ull; p.out_dists[o] = __uint_as_float(0x7fc00000u); }
    }
    if (tid == 0) {
        p.out_counts[q] = nout;
        if (sAnyNan) atomicOr(p.status, ST_NAN);
        p.cert[q] = sNanKey ? 0u : 1u;                           // (a truncated list is flagged by the select: overflow)
    }
}
void launch_rerank_all(const RerankParams& p, uint32_t nq, hipStream_t s) {
    if (!nq) return;
    RerankParams q = p;
    uint32_t dimp = (p.dim + 3) & ~3u;
    q.lds_row_stride = dimp + ((dimp % 8 == 0) ? 4 : 0);
    uint32_t chunk = (uint32_t)std::min<size_t>(64, (150 * 1024) / ((size_t)q.lds_row_stride * 4));
    chunk = chunk > 1 ? chunk - 1 : 1;
    q.lds_chunk = chunk;
    size_t lds = (size_t)(chunk + 1) * q.lds_row_stride * 4;
    hipLaunchKernelGGL(rerank_all_kernel, dim3(nq), dim3(RR_THREADS), lds, s, q);
}

void launch_rerank(const RerankParams& p, uint32_t nq, hipStream_t s) {
    if (!nq) return;
    // LDS plan: the query row, then a slice area for the candidate rows of a round (rerank_kernel sizes its K slices to it)
    RerankParams q = p;
    if (q.kp > RR_MAX) q.kp = RR_MAX;
    if (q.kp_first == 0 || q.kp_first > q.kp) q.kp_first = q.kp;
    if (q.kp_step == 0) q.kp_step = 32;
    uint32_t dimp = (p.dim + 3) & ~3u;
    q.lds_row_stride = dimp + 4;                                  // floats of the query row (start of the slice area)
    const size_t avail = (size_t)150 * 1024 / 4 - q.lds_row_stride;            // dim <= 16384: at least 21 K floats = 512 rows of 40
    q.lds_chunk = (uint32_t)avail;
    size_t lds = ((size_t)q.lds_row_stride + q.lds_chunk) * 4;
    hipLaunchKernelGGL(rerank_kernel, dim3(nq), dim3(RR_THREADS), lds, s, q);
}

// ---------------------------------------------------------------------------------------------
// Compact re-run of uncertified queries by the next tier: gather their padded rows and norms into a
// dense block, and scatter the block's results back to the batch positions.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_queries_kernel(const float* qp, const float* qnorm, uint32_t ld,
                                                             const uint32_t* qidx, uint32_t n, float* qp_out,
                                                             float* qnorm_out, float* thr_out) {
    const uint32_t j = blockIdx.x;                     // < n_pad: the padding rows are zero, their threshold -inf
    const bool real = j < n;
    const uint32_t q = real ? qidx[j] : 0u;
    for (uint32_t i = threadIdx.x; i < ld; i += blockDim.x) qp_out[(size_t)j * ld + i] = real ? qp[(size_t)q * ld + i] : 0.0f;
    if (threadIdx.x == 0) {
        qnorm_out[j] = real ? qnorm[q] : 0.0f;
        thr_out[j] = __uint_as_float(0xff800000u);
    }
}
void launch_gather_queries(const float* qp, const float* qnorm, uint32_t ld, const uint32_t* qidx, uint32_t n,
                           uint32_t n_pad, float* qp_out, float* qnorm_out, float* thr_out, hipStream_t s) {
    if (!n_pad) return;
    hipLaunchKernelGGL(gather_queries_kernel, dim3(n_pad), dim3(256), 0, s, qp, qnorm, ld, qidx, n, qp_out, qnorm_out,
                       thr_out);
}
__global__ __launch_bounds__(128) void scatter_results_kernel(const uint64_t* ids, const float* dists,
                                                              const uint32_t* counts, const uint32_t* qidx, uint32_t k,
                                                              uint64_t* out_ids, float* out_dists, uint32_t* out_counts) {
    const uint32_t j = blockIdx.x, q = qidx[j];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(size_t)q * k + i] = ids[(size_t)j * k + i];
        out_dists[(size_t)q * k + i] = dists[(size_t)j * k + i];
    }
    if (threadIdx.x == 0) out_counts[q] = counts[j];
}
void launch_scatter_results(const uint64_t* ids, const float* dists, const uint32_t* counts, const uint32_t* qidx,
                            uint32_t n, uint32_t k, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                            hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(scatter_results_kernel, dim3(n), dim3(128), 0, s, ids, dists, counts, qidx, k, out_ids, out_dists,
                       out_counts);
}

__global__ __launch_bounds__(128) void scatter_results_list_kernel(const uint64_t* ids, const float* dists, const uint32_t* counts,
                                                                   const uint32_t* src, const uint32_t* dst, uint32_t k,
                                                                   uint64_t* out_ids, float* out_dists, uint32_t* out_counts) {
    const uint32_t j = src[blockIdx.x], q = dst[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(size_t)q * k + i] = ids[(size_t)j * k + i];
        out_dists[(size_t)q * k + i] = dists[(size_t)j * k + i];
    }
    if (threadIdx.x == 0) out_counts[q] = counts[j];
}
void launch_scatter_results_list(const uint64_t* ids, const float* dists, const uint32_t* counts, const uint32_t* src,
                                 const uint32_t* dst, uint32_t n, uint32_t k, uint64_t* out_ids, float* out_dists,
                                 uint32_t* out_counts, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(scatter_results_list_kernel, dim3(n), dim3(128), 0, s, ids, dists, counts, src, dst, k, out_ids, out_dists,
                       out_counts);
}

// ---------------------------------------------------------------------------------------------
// Exact scan (fallback and large k): one thread per row, the reference's arithmetic, for
// one query.  Keys carry the id rank so that the select orders by (distance, id).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exact_scan_kernel(ExactScanParams p) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.n_rows) return;
    bool ok = p.rowmask ? ((p.rowmask[row >> 5] >> (row & 31)) & 1u) : true;
    uint64_t key = EMPTY_KEY;
    if (ok) {
        float dist = exact_distance(p.metric, p.q, p.rows + (size_t)row * p.ld, p.dim, p.qnorm[0], p.nd[row]);
        if (dist != dist) atomicOr(p.status, ST_NAN);
        uint32_t rk = p.idrank ? p.idrank[row] : row;
        key = ((uint64_t)f32_to_ordered(dist) << 32) | rk;
    }
    p.keys[row] = key;
}
void launch_exact_scan(const ExactScanParams& p, hipStream_t s) {
    if (!p.n_rows) return;
    hipLaunchKernelGGL(exact_scan_kernel, dim3((p.n_rows + 255) / 256), dim3(256), 0, s, p);
}

// The direct path of small indexes (SmallScanParams): blockIdx.y = query.  The workgroup stages the RAW query row in LDS (it may
// come from mapped host memory: one read over PCIe per workgroup), every thread computes the exact-order query norm it needs
// under Cosine from there (vector.rs:35-37: sqrt of the left fold of x*x; the same value in every thread), then one row per
// thread in the reference's operation order -- exact_distance, the function the re-rank and the exact scans use.
__global__ __launch_bounds__(256) void small_scan_kernel(SmallScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float sQ[];         // [round_up(dim, 4)]
    const uint32_t q = blockIdx.y;
    const float* src = p.q_in + (size_t)q * p.dim;
    for (uint32_t i = threadIdx.x; i < p.dim; i += blockDim.x) sQ[i] = src[i];
    __syncthreads();
    float qn = 0.0f;
    if (p.metric == COSINE) {
        qn = __builtin_sqrtf(fold_sq(sQ, p.dim));
        if (qn == 0.0f && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(p.status, ST_ZERO_QUERY);   // distance.rs:51-55
    }
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = row < p.n_rows && (p.rowmask ? ((p.rowmask[row >> 5] >> (row & 31)) & 1u) : true);
    uint64_t key = EMPTY_KEY;
    if (ok) {
        const float dist = exact_distance(p.metric, sQ, p.rows + (size_t)row * p.ld, p.dim, qn, p.nd[row]);
        if (dist != dist) atomicOr(p.status, ST_NAN);
        const uint32_t rk = p.idrank ? p.idrank[row] : row;
        key = ((uint64_t)f32_to_ordered(dist) << 32) | rk;
    }
    if (p.keep == 0) {
        if (row < p.n_rows) p.keys[(size_t)q * p.key_stride + row] = key;
        return;
    }
    // the workgroup's `keep` smallest keys, by counting (valid keys are distinct: the id rank is their low word)
    __shared__ __attribute__((aligned(16))) uint64_t sK[256];
    __shared__ uint32_t sValid;
    if (threadIdx.x == 0) sValid = 0;
    sK[threadIdx.x] = key;
    __syncthreads();
    const unsigned long long vb = __ballot(key != EMPTY_KEY);
    if ((threadIdx.x & 63) == 0 && vb) atomicAdd(&sValid, (uint32_t)__popcll(vb));
    uint32_t rank = 0;
    if (key != EMPTY_KEY) {
        for (uint32_t j = 0; j < 256; j += 2) {
            const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(sK + j);
            rank += (k2.x < key) + (k2.y < key);
        }
    }
    __syncthreads();
    uint64_t* out = p.keys + (size_t)q * p.key_stride + (size_t)blockIdx.x * p.keep;
    if (key != EMPTY_KEY && rank < p.keep) out[rank] = key;
    const uint32_t nv = sValid < p.keep ? sValid : p.keep;
    if (threadIdx.x >= nv && threadIdx.x < p.keep) out[threadIdx.x] = EMPTY_KEY;
}
uint32_t small_scan_groups(uint32_t n_rows) { return (n_rows + 255) / 256; }
void launch_small_scan(const SmallScanParams& p, hipStream_t s) {
    if (!p.n_rows || !p.nq) return;
    hipLaunchKernelGGL(small_scan_kernel, dim3((p.n_rows + 255) / 256, p.nq), dim3(256), (size_t)((p.dim + 3) & ~3u) * sizeof(float), s, p);
}

// One pass over the rows for up to 8 queries: one thread per row, the row is read once (16 floats at a
// time) and folded against every query in the reference's order (independent chains -> ILP).
__global__ __launch_bounds__(256) void exact_multi_kernel(ExactMultiParams p) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.n_rows) return;
    if (p.rowmask && !((p.rowmask[row >> 5] >> (row & 31)) & 1u)) return;
    const float* x = p.rows + (size_t)row * p.ld;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.0f;
    const uint32_t d = p.dim;
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < (int)p.nqf) {                                           // wave-uniform
                const float* q = p.qp + (size_t)p.qidx[j] * p.ld + i;       // uniform address: scalar loads
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(q + 4 * u);
                    if (p.metric == EUCLID) {
                        float t;
                        t = __fsub_rn(a.x, v[u].x); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.y, v[u].y); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.z, v[u].z); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.w, v[u].w); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                    } else {
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.x, v[u].x));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.y, v[u].y));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.z, v[u].z));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.w, v[u].w));
                    }
                }
            }
        }
    }
    for (; i < d; ++i) {
        const float xv = x[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < (int)p.nqf) {
                const float a = p.qp[(size_t)p.qidx[j] * p.ld + i];
                if (p.metric == EUCLID) { float t = __fsub_rn(a, xv); s[j] = __fadd_rn(s[j], __fmul_rn(t, t)); }
                else s[j] = __fadd_rn(s[j], __fmul_rn(a, xv));
            }
        }
    }
    const float xn = p.nd[row];
    const uint32_t rk = p.idrank ? p.idrank[row] : row;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < (int)p.nqf) {
            const uint32_t qi = p.qidx[j];
            float dist;
            if (p.metric == EUCLID) dist = __builtin_sqrtf(s[j]);
            else if (p.metric == DOT) dist = -s[j];
            else {
                float sim = __fdiv_rn(s[j], __fmul_rn(p.qnorm[qi], xn));
                if (sim < -1.0f) sim = -1.0f;
                if (sim > 1.0f) sim = 1.0f;
                dist = __fsub_rn(1.0f, sim);
            }
            if (dist != dist) atomicOr(p.status, ST_NAN);
            const float bound = (p.prev_counts[qi] == p.k) ? p.prev_dists[(size_t)qi * p.k + p.k - 1]
                                                           : __uint_as_float(0x7f800000u);
            if (!(dist > bound)) {                                          // ties with the bound are kept
                const uint32_t slot = atomicAdd(&p.cnt[j], 1u);
                if (slot < p.cap) p.keys[(size_t)j * p.cap + slot] = ((uint64_t)f32_to_ordered(dist) << 32) | rk;
            }
        }
    }
}
void launch_exact_multi(const ExactMultiParams& p, hipStream_t s) {
    if (!p.n_rows || !p.nqf) return;
    hipLaunchKernelGGL(exact_multi_kernel, dim3((p.n_rows + 255) / 256), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void emit_multi_kernel(EmitMultiParams p) {
    const uint32_t j = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t qi = p.qidx[j];
    uint32_t cnt = p.cnt[j];
    if (cnt > p.k) cnt = p.k;
    if (i == 0) p.out_count[qi] = cnt;
    if (i >= p.k) return;
    const size_t o = (size_t)qi * p.k + i;
    if (i < cnt) {
        uint64_t key = p.keys[(size_t)j * p.key_stride + i];
        uint32_t rk = (uint32_t)key;
        uint32_t row = p.rank2row ? p.rank2row[rk] : rk;
        p.out_ids[o] = p.row_ids[row];
        p.out_dists[o] = ordered_to_f32((uint32_t)(key >> 32));
    } else {
        p.out_ids[o] = ~0ull;
        p.out_dists[o] = __uint_as_float(0x7fc00000u);
    }
}
void launch_emit_multi(const EmitMultiParams& p, hipStream_t s) {
    if (!p.k || !p.nqf) return;
    hipLaunchKernelGGL(emit_multi_kernel, dim3((p.k + 255) / 256, p.nqf), dim3(256), 0, s, p);
}

__global__ __launch_bounds__(256) void emit_kernel(EmitParams p) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = *p.cnt;
    if (cnt > p.k) cnt = p.k;
    if (i == 0) *p.out_count = (p.accumulate ? *p.out_count : 0u) + cnt;
    if (i >= p.k) return;
    if (i < cnt) {
        uint64_t key = p.keys[i];
        uint32_t rk = (uint32_t)key;
        uint32_t row = p.rank2row ? p.rank2row[rk] : rk;
        p.out_ids[i] = p.row_ids[row];
        p.out_dists[i] = ordered_to_f32((uint32_t)(key >> 32));
    } else {
        p.out_ids[i] = ~0ull;
        p.out_dists[i] = __uint_as_float(0x7fc00000u);
    }
}
void launch_emit(const EmitParams& p, hipStream_t s) {
    if (!p.k) return;
    hipLaunchKernelGGL(emit_kernel, dim3((p.k + 255) / 256), dim3(256), 0, s, p);
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU exchange: merge nparts sorted partial top-k lists per query by (distance, id).
// One workgroup per query; nparts*k <= 2048 candidates sorted in LDS.
// ---------------------------------------------------------------------------------------------
// DistanceMetric::distance for explicit (query, stored row) pairs: what an HNSW search_layer asks for at
// src/hnsw/graph.rs:155 and :182 (at most 2m = 32 neighbours per expansion), batched over many queries.
__global__ __launch_bounds__(256) void pair_distances_kernel(PairDistParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_pairs) return;
    const uint32_t row = p.pair_row[i], q = p.pair_query[i];
    float d = __uint_as_float(0x7fc00000u);
    if (row != 0xffffffffu) {
        const float qn = p.qnorm[q], xn = p.nd[row];
        if (p.metric == COSINE && (qn == 0.0f || xn == 0.0f)) atomicOr(p.status, ST_ZERO_QUERY);
        d = exact_distance(p.metric, p.qp + (size_t)q * p.ld, p.rows + (size_t)row * p.ld, p.dim, qn, xn);
        if (d != d) atomicOr(p.status, ST_NAN);
    }
    p.out[i] = d;
}
void launch_pair_distances(const PairDistParams& p, hipStream_t s) {
    if (!p.n_pairs) return;
    hipLaunchKernelGGL(pair_distances_kernel, dim3((p.n_pairs + 255) / 256), dim3(256), 0, s, p);
}

// HNSW hooks (vdb_internal.h): exact reference distances of explicit (query, row) or (row, row) pairs, or of one
// query against rows [0, n); inputs and outputs may live in mapped host memory.
__global__ __launch_bounds__(256) void pair_eval_kernel(PairEvalParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const float* x; const float* y; float xn, yn;
    if (p.mode == 1) {
        const uint32_t ra = p.a[i], rb = p.b[i];
        x = p.rows + (size_t)ra * p.ld; xn = p.nd[ra];
        y = p.rows + (size_t)rb * p.ld; yn = p.nd[rb];
    } else {
        const uint32_t q = p.mode == 0 ? p.a[i] : p.q0, r = p.mode == 0 ? p.b[i] : i;
        x = p.qp + (size_t)q * p.ld; xn = p.qnorm[q];
        y = p.rows + (size_t)r * p.ld; yn = p.nd[r];
    }
    float d;
    if (p.metric == COSINE && (xn == 0.0f || yn == 0.0f)) d = __uint_as_float(p.mark);
    else d = exact_distance(p.metric, x, y, p.dim, xn, yn);
    p.out[i] = d;
}
void launch_pair_eval(const PairEvalParams& p, hipStream_t s) {
    if (!p.n) return;
    hipLaunchKernelGGL(pair_eval_kernel, dim3((p.n + 255) / 256), dim3(256), 0, s, p);
}

// One pass over rows [0, n_scan) for up to 16 query rows: one thread per row, the row read once (16 floats at a time)
// and folded against every query in the reference's order (independent chains -> ILP); the query elements come through
// scalar loads (uniform addresses).  The all-pairs scan the batched HNSW build amortises over a chunk of inserts.
template <int NQ>
__global__ __launch_bounds__(256) void scan_rows_kernel(ScanRowsParams p) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= p.n_scan) return;
    const float* x = p.rows + (size_t)row * p.ld;
    float s[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) s[j] = 0.0f;
    const uint32_t d = p.dim;
    uint32_t i = 0;
    for (; i + 16 <= d; i += 16) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (j < (int)p.nq) {                                            // wave-uniform
                const float* q = p.rows + (size_t)p.qrow[j] * p.ld + i;     // uniform address: scalar loads
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(q + 4 * u);
                    if (p.metric == EUCLID) {
                        float t;
                        t = __fsub_rn(a.x, v[u].x); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.y, v[u].y); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.z, v[u].z); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                        t = __fsub_rn(a.w, v[u].w); s[j] = __fadd_rn(s[j], __fmul_rn(t, t));
                    } else {
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.x, v[u].x));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.y, v[u].y));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.z, v[u].z));
                        s[j] = __fadd_rn(s[j], __fmul_rn(a.w, v[u].w));
                    }
                }
            }
        }
    }
    for (; i < d; ++i) {
        const float xv = x[i];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (j < (int)p.nq) {
                const float a = p.rows[(size_t)p.qrow[j] * p.ld + i];
                if (p.metric == EUCLID) { float t = __fsub_rn(a, xv); s[j] = __fadd_rn(s[j], __fmul_rn(t, t)); }
                else s[j] = __fadd_rn(s[j], __fmul_rn(a, xv));
            }
        }
    }
    const float xn = p.nd[row];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        if (j < (int)p.nq) {
            float dist;
            if (p.metric == EUCLID) dist = __builtin_sqrtf(s[j]);
            else if (p.metric == DOT) dist = -s[j];
            else {
                const float qn = p.nd[p.qrow[j]];
                if (qn == 0.0f || xn == 0.0f) dist = __uint_as_float(p.mark);
                else {
                    float sim = __fdiv_rn(s[j], __fmul_rn(qn, xn));          // norm1 * norm2 with the QUERY's norm first (distance.rs:58)
                    if (sim < -1.0f) sim = -1.0f;
                    if (sim > 1.0f) sim = 1.0f;
                    dist = __fsub_rn(1.0f, sim);
                }
            }
            p.out[(size_t)j * p.ldm + row] = dist;
        }
    }
}
void launch_scan_rows(const ScanRowsParams& p, hipStream_t s) {
    if (!p.n_scan || !p.nq) return;
    const dim3 grid((p.n_scan + 255) / 256), block(256);
    if (p.nq <= 4) hipLaunchKernelGGL(scan_rows_kernel<4>, grid, block, 0, s, p);
    else if (p.nq <= 8) hipLaunchKernelGGL(scan_rows_kernel<8>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(scan_rows_kernel<16>, grid, block, 0, s, p);
}

__global__ void write_code_kernel(const uint32_t* flags, int32_t* code) { *code = (flags[0] | flags[1]) ? 100 : 0; }
void launch_write_code(const uint32_t* flags, int32_t* code, hipStream_t s) {
    hipLaunchKernelGGL(write_code_kernel, dim3(1), dim3(1), 0, s, flags, code);
}

constexpr uint32_t MERGE_MAX = 2048;
// part p's arrays start at ids + p*ids_stride, dists + p*dists_stride, counts + p*counts_stride (element units),
// so both the plain [nparts][nq][k] layout and the packed all-gather buffer of sharded.py can be merged in place
__global__ __launch_bounds__(256) void merge_parts_kernel(const uint64_t* ids, const float* dists,
                                                          const uint32_t* counts, size_t ids_stride,
                                                          size_t dists_stride, size_t counts_stride,
                                                          const uint32_t* status, size_t status_stride,
                                                          uint32_t nparts, uint32_t nq,
                                                          uint32_t k, uint64_t* out_ids, float* out_dists,
                                                          uint32_t* out_counts, uint32_t* out_status) {
    __shared__ uint32_t sD[MERGE_MAX];
    __shared__ uint64_t sI[MERGE_MAX];
    __shared__ uint32_t sTotal;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t total = nparts * k;
    uint32_t P = 2;
    while (P < total) P <<= 1;
    if (tid == 0) sTotal = 0;
    if (q == 0 && tid == 0 && out_status) {
        uint32_t worst = 0;
        for (uint32_t part = 0; part < nparts; ++part) { uint32_t v = status[part * status_stride]; worst = v > worst ? v : worst; }
        *out_status = worst;
    }
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < P; i += 256) {
        uint32_t od = 0xffffffffu;
        uint64_t id = ~0ull;
        if (i < total) {
            uint32_t part = i / k, j = i - part * k;
            if (j < counts[part * counts_stride + q]) {
                od = f32_to_ordered(dists[part * dists_stride + (size_t)q * k + j]);
                id = ids[part * ids_stride + (size_t)q * k + j];
                ++mine;
            }
        }
        sD[i] = od;
        sI[i] = id;
    }
    if (mine) atomicAdd(&sTotal, mine);
    __syncthreads();
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < P / 2; t += 256) {
                uint32_t lo = 2 * t - (t & (stride - 1));
                uint32_t hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint32_t da = sD[lo], db = sD[hi];
                uint64_t ia = sI[lo], ib = sI[hi];
                bool gt = da > db || (da == db && ia > ib);
                if (gt == up) { sD[lo] = db; sD[hi] = da; sI[lo] = ib; sI[hi] = ia; }
            }
            __syncthreads();
        }
    }
    const uint32_t nout = sTotal < k ? sTotal : k;
    for (uint32_t i = tid; i < k; i += 256) {
        size_t o = (size_t)q * k + i;
        if (i < nout) { out_ids[o] = sI[i]; out_dists[o] = ordered_to_f32(sD[i]); }
        else { out_ids[o] = ~0ull; out_dists[o] = __uint_as_float(0x7fc00000u); }
    }
    if (tid == 0) out_counts[q] = nout;
}
void launch_merge_parts(const uint64_t* ids, const float* dists, const uint32_t* counts, uint32_t nparts,
                        uint32_t nq, uint32_t k, uint64_t* out_ids, float* out_dists, uint32_t* out_counts,
                        hipStream_t s) {
    if (!nq || !k) return;
    hipLaunchKernelGGL(merge_parts_kernel, dim3(nq), dim3(256), 0, s, ids, dists, counts, (size_t)nq * k, (size_t)nq * k,
                       (size_t)nq, nullptr, (size_t)0, nparts, nq, k, out_ids, out_dists, out_counts, nullptr);
}
// packed layout of one part (int32 words): ids int64[nq*k] | dists f32[nq*k] | counts i32[nq] | status i32 | pad
void launch_merge_packed(const int32_t* packed, size_t words_per_part, uint32_t nparts, uint32_t nq, uint32_t k,
                         uint64_t* out_ids, float* out_dists, uint32_t* out_counts, uint32_t* out_status,
                         hipStream_t s) {
    if (!nq || !k) return;
    const size_t nk = (size_t)nq * k;
    hipLaunchKernelGGL(merge_parts_kernel, dim3(nq), dim3(256), 0, s, reinterpret_cast<const uint64_t*>(packed),
                       reinterpret_cast<const float*>(packed + 2 * nk), reinterpret_cast<const uint32_t*>(packed + 3 * nk),
                       words_per_part / 2, words_per_part, words_per_part,
                       reinterpret_cast<const uint32_t*>(packed + 3 * nk + nq), words_per_part, nparts, nq, k, out_ids,
                       out_dists, out_counts, out_status);
}

}  // namespace vdb
