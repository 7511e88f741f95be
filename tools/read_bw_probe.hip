// read_bw_probe.hip -- what a READ-ONLY stream reaches on this chip (the screening kernel's floor), against the 6.3 TB/s a
// copy reaches: grid-stride float4 loads of a 3 GB buffer, several unroll depths / workgroup counts, plain and `nt`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const f4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int U, bool NT> void run(const f4* d, size_t n4, float* o, int wgs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rd<U, NT>), dim3(wgs), dim3(256), 0, 0, d, n4, o);
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((rd<U, NT>), dim3(wgs), dim3(256), 0, 0, d, n4, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("U=%d nt=%d wgs=%5d: %.3f ms  %.2f TB/s\n", U, (int)NT, wgs, ms, n4 * 16.0 / ms / 1e9);
}
int main() {
    const size_t bytes = 3072ull * 1000 * 1000, n4 = bytes / 16;
    f4* d; float* o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    for (int wgs : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) { run<4, false>(d, n4, o, wgs); run<8, false>(d, n4, o, wgs); run<8, true>(d, n4, o, wgs); run<16, true>(d, n4, o, wgs); }
    return 0;
}
