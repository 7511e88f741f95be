// read_pattern_probe.hip -- which part of the screening kernel's ACCESS PATTERN costs HBM bandwidth?  A read-only stream
// reaches 7.5 TB/s on this chip (tools/read_bw_probe.hip); the kernel's DMA-only loop 6.3.  Variants, all reading the same
// 1M x 768 f32 matrix (3.07 GB) once with 256 persistent workgroups of 512 threads, 16 bytes per lane and request:
//   A  contiguous range of rows per workgroup, rows read front to back (each lane-group streams 128 B lines in order)
//   B  the kernel's order: tiles of 256 rows, 24 K-stages per tile, per stage 128 B of each of the tile's 256 rows
//   C  as B but 256 B of each row per stage (12 double stages)
//   D  as B with tiles dealt round-robin to the workgroups (wg w takes tiles w, w+256, ...) instead of contiguous ranges
//   E  as C with round-robin tiles
// UNROLL stage-loads are kept in flight per lane before they are consumed (registers, not LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 1000000, LD = 768, NT = 512, TR = 256;
template <int SEG /*bytes of a row per stage*/, bool RR, int UNROLL>
__global__ __launch_bounds__(NT) void pat(const char* __restrict__ rows, float* out) {
    const int nblk = (N + TR - 1) / TR, nwg = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    constexpr int LPR = SEG / 16;                       // lanes per row segment
    constexpr int RPI = NT / LPR;                       // rows per instruction of the whole workgroup
    constexpr int IPS = TR / RPI;                       // instructions per stage and thread
    constexpr int KS = LD * 4 / SEG;
    const int rsub = tid / LPR, coff = (tid % LPR) * 16;
    float acc = 0.f;
    int t0, t1, tstep;
    if (RR) { t0 = w; t1 = nblk; tstep = nwg; } else { t0 = (int)((long)w * nblk / nwg); t1 = (int)((long)(w + 1) * nblk / nwg); tstep = 1; }
    for (int t = t0; t < t1; t += tstep) {
        const char* base = rows + (size_t)t * TR * LD * 4;
        for (int ks = 0; ks < KS; ks += UNROLL) {
            f4 v[UNROLL][IPS];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < IPS; ++i) {
                    int r = i * RPI + rsub; if (t * TR + r >= N) r = 0;
                    v[u][i] = __builtin_nontemporal_load((const f4*)(base + (size_t)r * LD * 4 + (size_t)(ks + u) * SEG + coff));
                }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < IPS; ++i) acc += v[u][i].x + v[u][i].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(NT) void seq(const char* __restrict__ rows, float* out) {       // A
    const int nwg = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    const size_t total = (size_t)N * LD * 4, b0 = total / nwg * w / 8192 * 8192, b1 = (w + 1 == nwg) ? total : total / nwg * (w + 1) / 8192 * 8192;
    float acc = 0.f;
    for (size_t o = b0 + (size_t)tid * 16; o + 3 * NT * 16 < b1; o += 4 * NT * 16) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const f4*)(rows + o + (size_t)u * NT * 16));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// ---- the same order through the LDS-DMA path the screening kernel uses: 3-image ring, per stage and wave 4 row pieces
// (1 KB each: 8 rows x 128 B) and optionally 2 query pieces from an L2-resident image, a counted wait and ONE barrier per stage.
typedef __attribute__((address_space(3))) char* lds_ptr_t;
#define DMA(GP, LP, NTS) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " NTS \
        :: "s"((uint32_t)(uintptr_t)(lds_ptr_t)(LP)), "v"((const void*)(GP)) : "memory", "m0")
template <bool QUERIES, bool NTLOAD, bool RR, int BARRIER /*0 none, 1 per stage*/>
__global__ __launch_bounds__(NT) void dma(const char* __restrict__ rows, const char* __restrict__ qimg, float* out) {
    extern __shared__ char smem[];
    const int nblk = (N + TR - 1) / TR, nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int t0, t1, tstep;
    if (RR) { t0 = wg; t1 = nblk; tstep = nwg; } else { t0 = (int)((long)wg * nblk / nwg); t1 = (int)((long)(wg + 1) * nblk / nwg); tstep = 1; }
    const int ntiles = (t1 - t0 + tstep - 1) / tstep, total = ntiles * 24;
    constexpr int IMG = 48 * 1024;
    auto issue = [&](int st) {
        const int t = t0 + (st / 24) * tstep, ks = st % 24;
        char* img = smem + (st % 3) * IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int r = 32 * w + 8 * j + (lane >> 3); if (t * TR + r >= N) r = 0;
            const char* gp = rows + ((size_t)t * TR + r) * LD * 4 + ks * 128 + (lane & 7) * 16;
            if (NTLOAD) DMA(gp, img + (4 * w + j) * 1024, "nt"); else DMA(gp, img + (4 * w + j) * 1024, "");
        }
        if (QUERIES) {
#pragma unroll
            for (int j = 0; j < 2; ++j) DMA(qimg + (size_t)ks * 16384 + (2 * w + j) * 1024 + lane * 16, img + 32768 + (2 * w + j) * 1024, "");
        }
    };
    constexpr int PER = QUERIES ? 6 : 4;
    issue(0); issue(1); issue(2);
    for (int st = 0; st < total; ++st) {
        // stage st must have landed: at most the pieces of st+1 and st+2 outstanding
        if (st + 2 < total) { if (PER == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (st + 3 < total) issue(st + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[tid * 16] == 77 && smem[tid * 16 + 40000] == 78) out[0] = 1.f;
}

// ---- bf16 shadow rows (1536 B per row): 4 waves per workgroup, K = 64 per stage (128 B per row and stage, 12 stages per tile),
// row ring of DEPTH x 32 KB, query image 2 x 16 KB per stage from L2 into a 3 x 16 KB ring
template <bool QUERIES, int DEPTH>
__global__ __launch_bounds__(256) void dma16(const char* __restrict__ rows, const char* __restrict__ qimg, float* out) {
    extern __shared__ char smem[];
    constexpr int LDB = 1536, KS16 = 12;
    const int nblk = (N + TR - 1) / TR, nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t0 = (int)((long)wg * nblk / nwg), t1 = (int)((long)(wg + 1) * nblk / nwg);
    const int total = (t1 - t0) * KS16;
    auto issue = [&](int st) {
        const int t = t0 + st / KS16, ks = st % KS16;
        char* img = smem + (st % DEPTH) * 32768;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = 64 * w + 8 * j + (lane >> 3); if (t * TR + r >= N) r = 0;
            DMA(rows + ((size_t)t * TR + r) * LDB + ks * 128 + (lane & 7) * 16, img + (8 * w + j) * 1024, "nt");
        }
        if (QUERIES) {
            char* qi = smem + DEPTH * 32768 + (st % 3) * 16384;      // (probe: both halves into the same slot; only the traffic matters)
#pragma unroll
            for (int j = 0; j < 8; ++j) DMA(qimg + (size_t)ks * 32768 + (8 * w + j) * 1024 + lane * 16, qi + ((8 * w + j) & 15) * 1024, "");
        }
    };
    constexpr int PER = QUERIES ? 16 : 8;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i);
    for (int st = 0; st < total; ++st) {
        if (st + DEPTH - 1 < total) {
            if (PER * (DEPTH - 1) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (PER * (DEPTH - 1) == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (PER * (DEPTH - 1) == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + DEPTH < total) issue(st + DEPTH);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[tid * 16] == 77 && smem[tid * 16 + 40000] == 78) out[0] = 1.f;
}
template <class F> void timeit(const char* name, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-58s %.3f ms  %.2f TB/s\n", name, ms, (double)N * LD * 4 / ms / 1e9);
}
int main() {
    char* d; float* o; const size_t bytes = ((size_t)N + 256) * LD * 4;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    timeit("A  contiguous range per wg, sequential", [&] { hipLaunchKernelGGL(seq, dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("B  kernel order: 128 B per row and stage, ranges, 2 deep", [&] { hipLaunchKernelGGL((pat<128, false, 2>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("B4 kernel order: 128 B per row and stage, ranges, 4 deep", [&] { hipLaunchKernelGGL((pat<128, false, 4>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("C  256 B per row and stage, ranges, 2 deep", [&] { hipLaunchKernelGGL((pat<256, false, 2>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("D  128 B per row and stage, round-robin tiles, 2 deep", [&] { hipLaunchKernelGGL((pat<128, true, 2>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("D4 128 B per row and stage, round-robin tiles, 4 deep", [&] { hipLaunchKernelGGL((pat<128, true, 4>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("E  256 B per row and stage, round-robin tiles, 2 deep", [&] { hipLaunchKernelGGL((pat<256, true, 2>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("F  512 B per row and stage, ranges, 1 deep", [&] { hipLaunchKernelGGL((pat<512, false, 1>), dim3(256), dim3(NT), 0, 0, d, o); });
    timeit("G  3072 B (whole row) per stage, ranges, 1 deep", [&] { hipLaunchKernelGGL((pat<3072, false, 1>), dim3(256), dim3(NT), 0, 0, d, o); });
    char* q; hipMalloc(&q, 24 * 16384); hipMemset(q, 0, 24 * 16384);
    const int SM = 3 * 48 * 1024;
#define RUN(NAME, ...) { auto k = dma<__VA_ARGS__>; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SM); \
        timeit(NAME, [&] { hipLaunchKernelGGL(k, dim3(256), dim3(NT), SM, 0, d, q, o); }); }
    RUN("H  LDS-DMA rows only, nt, ranges, barrier per stage", false, true, false, 1)
    RUN("H0 LDS-DMA rows only, nt, ranges, no barrier", false, true, false, 0)
    RUN("H1 LDS-DMA rows only, plain, ranges, barrier per stage", false, false, false, 1)
    RUN("I  LDS-DMA rows nt + query image, ranges, barrier", true, true, false, 1)
    RUN("I0 LDS-DMA rows nt + query image, ranges, no barrier", true, true, false, 0)
    RUN("I1 LDS-DMA rows plain + query image, ranges, barrier", true, false, false, 1)
    RUN("K  LDS-DMA rows nt + query image, round-robin tiles, barrier", true, true, true, 1)
    char* q2; hipMalloc(&q2, 12 * 32768); hipMemset(q2, 0, 12 * 32768);
#define RUN16(NAME, Q, DEPTH) { auto k = dma16<Q, DEPTH>; const int sm = DEPTH * 32768 + 3 * 16384; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
        timeit(NAME, [&] { hipLaunchKernelGGL(k, dim3(256), dim3(256), sm, 0, d, q2, o); }); }
    printf("bf16 shadow rows: 1.536 GB per launch (the TB/s column is for 3.07 GB: halve it)\n");
    RUN16("S2  bf16 rows only, ring 2", false, 2)
    RUN16("S3  bf16 rows only, ring 3", false, 3)
    RUN16("S3q bf16 rows + query image, ring 3", true, 3)
    RUN16("S2q bf16 rows + query image, ring 2", true, 2)
    return 0;
}
