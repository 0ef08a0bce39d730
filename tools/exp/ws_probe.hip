// ws_probe.hip — does a store stream issued by "helper" waves run beside an fp32-MFMA loop on the SAME CU at full
// rate?  (Design question for the wave-specialised policy+step+obs kernel.)  512-thread workgroups, one per CU:
// waves 0-3 run back-to-back v_mfma_f32_32x32x2_f32 (8 independent accumulators, like the gate GEMM), waves 4-7 write
// their share of a 1.19 GB buffer with 16-byte stores (1 KiB per wave instruction).  Three timings: MFMA only, stores
// only, both.   hipcc --offload-arch=gfx950 -O3 ws_probe.hip -o ws_probe && ./ws_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bits: 1 = store waves raise their priority (s_setprio 3); 2 = roles split by WORKGROUP parity instead of by wave
// (even workgroups: 8 MFMA-less... see below); 4 = the MFMA waves themselves issue the stores (one 1 KiB store per
// `every` MFMAs), no helper waves; 8 = helper waves stream LOADS instead of stores
__global__ __launch_bounds__(512, 1) void probe(float* out, size_t floats_total, int mfma_blocks, int do_mfma, int do_store,
                                                int store_waves, float* sink, int mode, int every)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (mode & 4) {   // same-wave interleave: waves 0-3 only
        if (w >= 4) return;
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        float a = (float)lane * 1e-3f, b = (float)w * 1e-3f;
        const size_t per_cu = floats_total / 4 / gridDim.x;
        f32x4* base = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.x * per_cu;
        size_t q = (size_t)w * 64 + lane;
        const f32x4 v = { 1.f, 0.f, 0.f, (float)lane };
        for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    if (((r * 8 + i) % every) == 0 && q < per_cu) {
                        base[q] = v;
                        q += 256;
                    }
                }
            }
        }
        for (; q < per_cu; q += 256) base[q] = v;
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    const bool mfma_role = (mode & 2) ? ((blockIdx.x & 1) == 0) : (w < 4);
    if (mfma_role) {
        if (!do_mfma) return;
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        float a = (float)lane * 1e-3f, b = (float)w * 1e-3f;
        if (mode & 4096) {        // bf16 MFMA stream (v_mfma_f32_32x32x16_bf16, 32 cycles each): twice the count, same duration
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ua = { 0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u };
            const u32x4 ub = { 0x3c003c00u + (unsigned)w, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u };
            const bf16x8 xa = __builtin_bit_cast(bf16x8, ua), xb = __builtin_bit_cast(bf16x8, ub);
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[i], 0, 0, 0);
            }
        } else if (mode & 16) {          // accumulators pinned to AGPRs
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            }
        } else if (mode & 32) {   // 16x16x4 (32-cycle) form, 16 independent accumulators of 4 registers
            typedef float f32x4a __attribute__((ext_vector_type(4)));
            f32x4a ac[16];
            for (int i = 0; i < 16; ++i) ac[i] = f32x4a{ 0.f, 0.f, 0.f, 0.f };
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ac[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, ac[i], 0, 0, 0);
            }
            for (int i = 0; i < 16; ++i) acc[0][0] += ac[i][0];
        } else if (mode & 256) {  // the MFMA wave idles 16 issue cycles after every MFMA
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 15");
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        } else if (mode & 1024) {  // the MFMA wave sleeps one quantum after EVERY MFMA (never waits at issue for the pipe)
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        } else if (mode & 2048) {  // 3 x s_nop 15 (48 idle issue cycles) after every MFMA
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        } else if (mode & 512) {  // the MFMA wave sleeps one quantum (64 cycles) after every 8 MFMAs
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int it = 0; it < mfma_blocks; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.678f) sink[0] = s;
        if (threadIdx.x == 0 && blockIdx.x < 256) {   // shader cycles and 100 MHz ticks spent in the MFMA loop
            reinterpret_cast<unsigned long long*>(sink)[2 + 2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
            reinterpret_cast<unsigned long long*>(sink)[3 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - r0;
        }
    } else {
        int sw = w - 4, nsw = store_waves;
        if (mode & 2) { sw = w; nsw = 8; }
        if (!do_store || sw < 0 || sw >= nsw) return;
        if (mode & 1) __builtin_amdgcn_s_setprio(3);
        // this CU's contiguous share, split between its store waves in 1 KiB (= one wave store) units
        const size_t q_total = floats_total / 4;                       // float4s
        const size_t nshare = (mode & 2) ? gridDim.x / 2 : gridDim.x;
        const size_t per_cu = q_total / nshare;
        f32x4* base = reinterpret_cast<f32x4*>(out) + (size_t)((mode & 2) ? blockIdx.x / 2 : blockIdx.x) * per_cu;
        const f32x4 v = { 1.f, 0.f, 0.f, (float)lane };
        if (mode & 64) {          // helper = pure VALU work: `every` x 1024 dependent-free FMAs per lane
            float x0 = (float)lane, x1 = 1.f, x2 = 2.f, x3 = 3.f;
            for (int it = 0; it < every * 256; ++it) {
                x0 = x0 * 1.0001f + 0.5f;
                x1 = x1 * 1.0001f + 0.5f;
                x2 = x2 * 1.0001f + 0.5f;
                x3 = x3 * 1.0001f + 0.5f;
            }
            if (x0 + x1 + x2 + x3 == 12345.678f) sink[1] = x0;
        } else if (mode & 128) {  // helper = dependent L2 loads (pointer chase over a small table), `every` x 64 hops
            const int* tab = reinterpret_cast<const int*>(out);
            int idx = lane;
            for (int it = 0; it < every * 64; ++it) idx = tab[idx & 16383] + lane;
            if (idx == 123456789) sink[1] = (float)idx;
        } else if (mode & 8192) {   // round 4: a store stream WITHOUT vector ALU work (descriptor + scalar offset), non-temporal
            typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
            const unsigned int mine = (unsigned int)(per_cu * 16);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(base), 0, mine, 0x00020000);
            u32x4s z = { 0u, 0u, 0u, 0u };
            asm volatile("" : "+v"(z));
            const int voff = lane * 16;
            int so = __builtin_amdgcn_readfirstlane(sw) * 1024;
            const int step = nsw * 1024;
#pragma unroll 1
            for (; so < (int)mine; so += step) __builtin_amdgcn_raw_buffer_store_b128(z, r, voff, so, 2);
        } else if (mode & 8) {
            f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
            for (size_t q = (size_t)sw * 64 + lane; q < per_cu; q += (size_t)nsw * 64) acc += base[q];
            if (acc.x == 12345.678f) sink[1] = acc.y;
        } else {
            for (size_t q = (size_t)sw * 64 + lane; q < per_cu; q += (size_t)nsw * 64) base[q] = v;
        }
    }
}

int main(int argc, char** argv)
{
    const size_t bytes = 1191444480;   // PP-hard obs of 8192 envs
    float *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 8192);
    hipMemset(sink, 0, 8192);
    hipMemset(buf, 0, 65536 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 512;            // x 32 MFMAs = 16384 per wave = 1.05 M cycles ~ 0.44 ms at 2.4 GHz
    struct { const char* name; int m, s, sw, mode, every; } cfg[] = {
        { "mfma only", 1, 0, 4, 0, 1 },
        { "stores only (4 waves/CU)", 0, 1, 4, 0, 1 },
        { "both (4 store waves)", 1, 1, 4, 0, 1 },
        { "both, store waves setprio 3", 1, 1, 4, 1, 1 },
        { "both, 1 store wave setprio 3", 1, 1, 1, 1, 1 },
        { "loads only (4 waves/CU)", 0, 1, 4, 8, 1 },
        { "both, LOAD stream (4 waves)", 1, 1, 4, 8, 1 },
        { "both, LOAD stream setprio 3", 1, 1, 4, 9, 1 },
        { "both, AGPR accumulators", 1, 1, 4, 16, 1 },
        { "mfma only, AGPR acc", 1, 0, 4, 16, 1 },
        { "mfma only, 16x16x4", 1, 0, 4, 32, 1 },
        { "both, 16x16x4", 1, 1, 4, 32, 1 },
        { "both, VALU helper setprio 3", 1, 1, 4, 64 + 1, 256 },
        { "mfma only, s_nop 15 per mfma", 1, 0, 4, 256, 1 },
        { "both, VALU helper, s_nop 15/mfma", 1, 1, 4, 64 + 256, 256 },
        { "both, VALU prio3, s_nop 15/mfma", 1, 1, 4, 64 + 256 + 1, 256 },
        { "both, stores, s_nop 15/mfma", 1, 1, 4, 256, 1 },
        { "mfma only, s_sleep 1 per mfma", 1, 0, 4, 1024, 1 },
        { "both, VALU helper, sleep/1", 1, 1, 4, 64 + 1024, 256 },
        { "both, stores, sleep/1", 1, 1, 4, 1024, 1 },
        { "mfma only, 3 s_nop15 per mfma", 1, 0, 4, 2048, 1 },
        { "both, VALU helper, 3 s_nop15", 1, 1, 4, 64 + 2048, 256 },
        { "both, stores, 3 s_nop15", 1, 1, 4, 2048, 1 },
        { "mfma only, s_sleep 1 per 8 mfma", 1, 0, 4, 512, 1 },
        { "both, VALU helper, sleep/8", 1, 1, 4, 64 + 512, 256 },
        { "both, stores, sleep/8", 1, 1, 4, 512, 1 },
        { "VALU helper only (4 waves)", 0, 1, 4, 64, 256 },
        { "both, VALU helper", 1, 1, 4, 64, 256 },
        { "both, VALU helper, AGPR acc", 1, 1, 4, 64 + 16, 256 },
        { "L2-chase helper only", 0, 1, 4, 128, 8 },
        { "both, L2-chase helper", 1, 1, 4, 128, 8 },
        { "both, L2-chase helper, AGPR", 1, 1, 4, 128 + 16, 8 },
        { "split by CU: mfma half only", 1, 0, 4, 2, 1 },
        { "split by CU: store half only", 0, 1, 4, 2, 1 },
        { "split by CU: both", 1, 1, 4, 2, 1 },
        { "same wave: store per 1 mfma", 1, 1, 4, 4, 1 },
        { "same wave: store per 2 mfma", 1, 1, 4, 4, 2 },
        { "same wave: store per 4 mfma", 1, 1, 4, 4, 4 },
        { "same wave: store per 8 mfma", 1, 1, 4, 4, 8 },
        // round 3: the same questions for a bf16 MFMA stream (flops column: fp32-MFMA-equivalent count x 1, i.e. 64 x 32-cycle
        // instructions per block instead of 32 x 64-cycle ones; multiply by 8 for bf16 flops)
        { "bf16 mfma only", 1, 0, 4, 4096, 1 },
        { "bf16 mfma + VALU helper", 1, 1, 4, 4096 + 64, 256 },
        { "bf16 mfma + VALU helper prio 3", 1, 1, 4, 4096 + 64 + 1, 256 },
        { "bf16 mfma + stores (4 waves)", 1, 1, 4, 4096, 1 },
        { "bf16 mfma + LOAD stream", 1, 1, 4, 4096 + 8, 1 },
        { "bf16 mfma + L2-chase helper", 1, 1, 4, 4096 + 128, 8 },
        // round 4: helper waves whose store loop holds no vector ALU instruction
        { "VALU-free stores only (4 waves)", 0, 1, 4, 8192, 1 },
        { "fp32 mfma + VALU-free stores", 1, 1, 4, 8192, 1 },
        { "bf16 mfma + VALU-free stores", 1, 1, 4, 4096 + 8192, 1 },
        { "fp32 mfma + VALU-free stores, 1 wave", 1, 1, 1, 8192, 1 },
        { "split by CU: VALU-free both", 1, 1, 4, 2 + 8192, 1 } };
    for (auto& c : cfg) {
        if (argc > 1 && !strstr(c.name, argv[1]) && strcmp(c.name, "mfma only") && strcmp(c.name, "bf16 mfma only")) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, buf, bytes / 4, blocks, c.m, c.s, c.sw, sink, c.mode, c.every);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double flops = ((c.mode & 2) ? 128.0 * 8 : 256.0 * 4) * blocks * 32 * 4096.0;
        printf("%-28s %.3f ms", c.name, best);
        if (c.m) printf("  mfma %.1f TFLOP/s", flops / (best * 1e-3) / 1e12);
        if (c.s) printf("  stores %.0f GB/s", bytes / (best * 1e-3) / 1e9);
        if (c.m && !(c.mode & 4)) {
            unsigned long long h[4];
            hipMemcpy(h, reinterpret_cast<unsigned long long*>(sink) + 2, sizeof(h), hipMemcpyDeviceToHost);
            printf("  | WG0 mfma loop: %.0f kcycles in %.3f ms = %.0f MHz shader clock", h[0] / 1e3, h[1] / 1e5,
                   h[0] / (h[1] / 100.0));
        }
        printf("\n");
    }
    return 0;
}
