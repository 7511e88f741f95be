// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 under the fused kernel's operand-feeding patterns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe   (diagnostic tool, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE, int NACC>
__global__ __launch_bounds__(512, 2) void probe(float* out, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a[NACC], b;
    b = make_float4(in[lane], in[lane + 1], in[lane + 2], in[lane + 3]);
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = make_float4(in[lane + i], in[lane + i + 4], in[lane + i + 8], in[lane + i + 12]);
    const float* lp = lds + (lane & 31) * 36 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {   // operands re-read from LDS each group of 4*NACC MFMAs (the fused kernel's pattern)
            b = *reinterpret_cast<const float4*>(lp + (it & 3) * 8);
#pragma unroll
            for (int i = 0; i < NACC; ++i) a[i] = *reinterpret_cast<const float4*>(lp + 1152 * i + (it & 3) * 8 + 1152 * NACC);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a[i].x, b.x, acc[i]);
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a[i].y, b.y, acc[i]);
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a[i].z, b.z, acc[i]);
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a[i].w, b.w, acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int NACC> void run(const char* name, int threads, int blocks_per_cu, float* out, float* in) {
    int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, NACC>), dim3(grid), dim3(threads), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = (double)threads / 64 * blocks_per_cu / 4;
    double mfma_per_simd = waves_per_simd * iters * 4.0 * NACC;
    double tf = 256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12;
    printf("%-44s threads=%d blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD; 64 cyc @2.4GHz = 26.7 ns)\n", name,
           threads, blocks_per_cu, ms, tf, ms * 1e6 / mfma_per_simd);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMalloc(&in, 4096 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 4>("regs only, 4 acc, 1 wave/SIMD", 256, 1, out, in);
    run<0, 4>("regs only, 4 acc, 2 waves/SIMD (1 WG)", 512, 1, out, in);
    run<0, 4>("regs only, 4 acc, 2 waves/SIMD (2 WG)", 256, 2, out, in);
    run<0, 8>("regs only, 8 acc, 1 wave/SIMD", 256, 1, out, in);
    run<1, 4>("LDS b128 operands, 4 acc, 1 wave/SIMD", 256, 1, out, in);
    run<1, 4>("LDS b128 operands, 4 acc, 2 waves/SIMD (1 WG)", 512, 1, out, in);
    run<1, 4>("LDS b128 operands, 4 acc, 2 waves/SIMD (2 WG)", 256, 2, out, in);
    run<1, 8>("LDS b128 operands, 8 acc, 1 wave/SIMD", 256, 1, out, in);
    return 0;
}
