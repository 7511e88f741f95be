// Microbenchmark: can f32 MFMA and packed f32 VALU FMAs (v_pk_fma_f32) run concurrently on one SIMD?
// Each loop iteration issues 4 MFMAs (32x32x2 f32) and NPK packed FMAs on independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe   (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NPK, int NMFMA>
__global__ __launch_bounds__(512, 2) void probe(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NACC = NPK > 0 ? NPK : 1;
    f32x2 v[NACC];
    for (int i = 0; i < NACC; ++i) v[i] = f32x2{in[lane + i], in[lane + i + 1]};
    float a0 = in[lane], a1 = in[lane + 1], a2 = in[lane + 2], a3 = in[lane + 3];
    f32x2 x = {in[lane + 4], in[lane + 5]}, y = {in[lane + 6], in[lane + 7]};
    for (int it = 0; it < iters; ++it) {
        if (NMFMA) {
            acc[0] = MFMA(a0, a1, acc[0]);
            acc[1] = MFMA(a1, a2, acc[1]);
            acc[2] = MFMA(a2, a3, acc[2]);
            acc[3] = MFMA(a3, a0, acc[3]);
        }
#pragma unroll
        for (int i = 0; i < NPK; ++i) v[i] = __builtin_elementwise_fma(v[i], x, y);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < NACC; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NPK, int NMFMA> void run(float* out, float* in) {
    int iters = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NPK, NMFMA>), dim3(256), dim3(512), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = 256.0 * 8;
    double mfma_flop = NMFMA ? waves * iters * 4.0 * 4096 : 0;
    double valu_flop = waves * iters * (double)NPK * 64 * 4;
    printf("MFMA x%d + pk_fma x%2d per iter: %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", NMFMA ? 4 : 0, NPK, ms,
           mfma_flop / ms / 1e9, valu_flop / ms / 1e9, (mfma_flop + valu_flop) / ms / 1e9);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&in, 4096 * 4);
    hipMemset(in, 0, 4096 * 4);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.001f * (i % 17) + 0.5f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0, 1>(out, in);
    run<16, 0>(out, in);
    run<16, 1>(out, in);
    run<32, 1>(out, in);
    run<48, 1>(out, in);
    run<64, 1>(out, in);
    return 0;
}
