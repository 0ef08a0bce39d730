// Experiment (not part of the product): write-stream throughput vs stores-per-thread U, fill-like geometry.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool STRIDED>
__global__ __launch_bounds__(256) void k(f32x4* out, long long n4) {
    // WG covers 256*U consecutive float4; STRIDED: thread's stores 256 apart (wave-contiguous 1 KB each),
    // else thread's U stores adjacent (64 B per lane)
    long long base = (long long)blockIdx.x * 256 * U;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long long g = STRIDED ? base + u * 256 + threadIdx.x : base + (long long)threadIdx.x * U + u;
        if (g < n4) out[g] = z;
    }
}
template <int U, bool S> float run(f32x4* d, long long n4) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned blocks = (unsigned)((n4 + 256 * U - 1) / (256 * U));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<U, S>), dim3(blocks), dim3(256), 0, 0, d, n4);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<U, S>), dim3(blocks), dim3(256), 0, 0, d, n4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
    const long long n4 = 1191444480ll / 16; f32x4* d; hipMalloc(&d, n4 * 16);
#define R(U, S) { float ms = run<U, S>(d, n4); printf("U=%d %s: %.3f ms %.0f GB/s\n", U, S ? "strided" : "adjacent", ms, n4 * 16 / ms / 1e6); }
    R(1, true) R(2, true) R(4, true) R(8, true) R(16, true) R(36, true) R(2, false) R(4, false) R(8, false)
    return 0;
}
