// ws_probe.hip — does a store stream issued by "helper" waves run beside an fp32-MFMA loop on the SAME CU at full
// rate?  (Design question for the wave-specialised policy+step+obs kernel.)  512-thread workgroups, one per CU:
// waves 0-3 run back-to-back v_mfma_f32_32x32x2_f32 (8 independent accumulators, like the gate GEMM), waves 4-7 write
// their share of a 1.19 GB buffer with 16-byte stores (1 KiB per wave instruction).  Three timings: MFMA only, stores
// only, both.   hipcc --offload-arch=gfx950 -O3 ws_probe.hip -o ws_probe && ./ws_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void probe(float* out, size_t floats_total, int mfma_blocks, int do_mfma, int do_store,
                                                int store_waves, float* sink)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w < 4) {
        if (!do_mfma) return;
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        float a = (float)lane * 1e-3f, b = (float)w * 1e-3f;
        for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.678f) sink[0] = s;
    } else {
        if (!do_store || w - 4 >= store_waves) return;
        // this CU's contiguous share, split between its store waves in 1 KiB (= one wave store) units
        const size_t q_total = floats_total / 4;                       // float4s
        const size_t per_cu = q_total / gridDim.x;
        f32x4* base = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.x * per_cu;
        const f32x4 v = { 1.f, 0.f, 0.f, (float)lane };
        for (size_t q = (size_t)(w - 4) * 64 + lane; q < per_cu; q += (size_t)store_waves * 64) base[q] = v;
    }
}

int main(int argc, char** argv)
{
    const size_t bytes = 1191444480;   // PP-hard obs of 8192 envs
    float *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 512;            // x 32 MFMAs = 16384 per wave = 1.05 M cycles ~ 0.44 ms at 2.4 GHz
    struct { const char* name; int m, s, sw; } cfg[] = { { "mfma only", 1, 0, 4 }, { "stores only (4 waves/CU)", 0, 1, 4 },
                                                          { "stores only (2 waves/CU)", 0, 1, 2 }, { "stores only (1 wave/CU)", 0, 1, 1 },
                                                          { "both (4 store waves)", 1, 1, 4 }, { "both (2 store waves)", 1, 1, 2 },
                                                          { "both (1 store wave)", 1, 1, 1 } };
    for (auto& c : cfg) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, buf, bytes / 4, blocks, c.m, c.s, c.sw, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double flops = 256.0 * 4 * blocks * 32 * 4096.0;
        printf("%-28s %.3f ms", c.name, best);
        if (c.m) printf("  mfma %.1f TFLOP/s", flops / (best * 1e-3) / 1e12);
        if (c.s) printf("  stores %.0f GB/s", bytes / (best * 1e-3) / 1e9);
        printf("\n");
    }
    return 0;
}
