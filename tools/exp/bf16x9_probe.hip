// bf16x9_probe.hip — what would the gate GEMM of policy_step_kernel cost on the bf16 matrix cores with EXACT products?
// (fp32 splits exactly into three bf16 terms a = a1 + a2 + a3; the nine products a_p * b_q are exact in
// fp32; v_mfma_f32_32x32x16_bf16 runs 16 k-steps in 32 cycles where v_mfma_f32_32x32x2_f32 runs 2 in 64: nine bf16 MFMAs
// per 16 k-steps = 1.78 x the fp32 rate, six (dropping the three terms below 2^-24 relative) = 2.67 x.)
// The probe runs C[R x 512] = A[R x 256] . W[256 x 512] (PP-hard's gate product, R = 81920) three ways on the same
// data — fp32 MFMA as shipped, bf16x9, bf16x6 — with the A tile of a workgroup in LDS (fp32, or three bf16 planes split
// by the workgroup itself from fp32 input: that cost is part of the timing) and the weights streamed from L2 in
// fragment order, and reports time and the error against an fp64 product of a row sample.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16x9_probe.hip -o bf16x9_probe && ./bf16x9_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 256, NC = 512, KB16 = K / 16, KB8 = K / 8;

__device__ __forceinline__ unsigned bf16_rne(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_f32(unsigned b) { return __builtin_bit_cast(float, b << 16); }
// a = a1 + a2 + a3 exactly (round-to-nearest-even splits; the residuals are exact in fp32)
__device__ __forceinline__ void split3(float a, unsigned& a1, unsigned& a2, unsigned& a3)
{
    a1 = bf16_rne(a);
    const float r1 = a - bf16_f32(a1);
    a2 = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(a2);
    a3 = bf16_rne(r2);
}

// ---- weights in fragment order ------------------------------------------------------------------------------------------
// fp32: Wq[kb8][ctile][lane] = float4 { W[8 kb8 + 4 lh + j][32 ctile + li] }, j = 0..3 (lane = 32 lh + li)
__global__ void pack_f32(const float* __restrict__ W, f32x4* __restrict__ Wq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KB8 * (NC / 32) * 64) return;
    const int lane = i & 63, ct = (i >> 6) % (NC / 32), kb = i / (64 * (NC / 32));
    const int li = lane & 31, lh = lane >> 5;
    f32x4 v;
    for (int j = 0; j < 4; ++j) v[j] = W[(size_t)(8 * kb + 4 * lh + j) * NC + 32 * ct + li];
    Wq[i] = v;
}
// bf16 planes: Wp[plane][kb16][ctile][lane] = 8 x bf16 { W_plane[16 kb16 + 8 lh + i][32 ctile + li] }, i = 0..7
__global__ void pack_bf16(const float* __restrict__ W, u32x4* __restrict__ Wp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = KB16 * (NC / 32) * 64;
    if (i >= per) return;
    const int lane = i & 63, ct = (i >> 6) % (NC / 32), kb = i / (64 * (NC / 32));
    const int li = lane & 31, lh = lane >> 5;
    unsigned p[3][8];
    for (int q = 0; q < 8; ++q) split3(W[(size_t)(16 * kb + 8 * lh + q) * NC + 32 * ct + li], p[0][q], p[1][q], p[2][q]);
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
        for (int d = 0; d < 4; ++d) v[d] = p[pl][2 * d] | (p[pl][2 * d + 1] << 16);
        Wp[(size_t)pl * per + i] = v;
    }
}

// C/D layout of the 32x32 MFMAs: register `reg` of lane (li, lh) <-> row (reg & 3) + 8 (reg >> 2) + 4 lh, column li
template <int RT>
__device__ __forceinline__ void store_c(float* __restrict__ C, size_t r0, int rows_left, int w, int li, int lh,
                                        const f32x16 (&acc)[RT][4])
{
    if (rows_left < 0) {   // timing without the 168 MB of output: one never-true store keeps the accumulators alive
        float s = 0.f;
        for (int rt = 0; rt < RT; ++rt)
            for (int g = 0; g < 4; ++g) s += acc[rt][g][0] + acc[rt][g][7];
        if (s == 123456.789f) C[0] = s;
        return;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (row < rows_left) C[(r0 + row) * NC + 32 * (4 * w + g) + li] = acc[rt][g][reg];
            }
}

// ---- fp32 MFMA, as shipped: A tile fp32 in LDS, 4 waves x 128 columns ----------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(256) void gemm_f32(const float* __restrict__ A, const f32x4* __restrict__ Wq, float* __restrict__ C,
                                                int R, int store)
{
    constexpr int RT = ROWS / 32, LDA = K + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * (K / 4); i += 256) {
        const int row = i / (K / 4), c4 = i % (K / 4);
        f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (r0 + row < (size_t)R) v = *reinterpret_cast<const f32x4*>(A + (r0 + row) * K + 4 * c4);
        *reinterpret_cast<f32x4*>(sm + row * LDA + 4 * c4) = v;
    }
    __syncthreads();
    f32x16 acc[RT][4];
    for (int rt = 0; rt < RT; ++rt)
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 16; ++i) acc[rt][g][i] = 0.f;
    const f32x4* wq = Wq + (size_t)(4 * w) * 64 + lane;
    f32x4 b[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b[0][g] = wq[g * 64];
#pragma unroll 2
    for (int kb = 0; kb < KB8; ++kb) {
        if (kb + 1 < KB8) {
#pragma unroll
            for (int g = 0; g < 4; ++g) b[(kb + 1) & 1][g] = wq[(size_t)(kb + 1) * (NC / 32) * 64 + g * 64];
        }
        f32x4 a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(sm + (32 * rt + li) * LDA + 8 * kb + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][j], b[kb & 1][g][j], acc[rt][g], 0, 0, 0);
    }
    store_c<RT>(C, r0, store ? (int)((size_t)R - r0 < (size_t)ROWS ? (size_t)R - r0 : (size_t)ROWS) : -1, w, li, lh, acc);
}

// ---- bf16 MFMA with exact products: NPROD = 9 (all) or 6 (terms >= 2^-16 relative) --------------------------------------
template <int ROWS, int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16(const float* __restrict__ A, const u32x4* __restrict__ Wp, float* __restrict__ C,
                                                 int R, int store)
{
    constexpr int RT = ROWS / 32, LDB = K + 8;                   // bf16 elements per LDS row (16-byte aligned rows)
    extern __shared__ __attribute__((aligned(16))) unsigned short sb[];   // [3][ROWS][LDB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    // the workgroup splits its own A tile: fp32 -> three bf16 planes in LDS
    for (int i = tid; i < ROWS * (K / 4); i += 256) {
        const int row = i / (K / 4), c4 = i % (K / 4);
        f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (r0 + row < (size_t)R) v = *reinterpret_cast<const f32x4*>(A + (r0 + row) * K + 4 * c4);
        unsigned p[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split3(v[q], p[0][q], p[1][q], p[2][q]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint2 o;
            o.x = p[pl][0] | (p[pl][1] << 16);
            o.y = p[pl][2] | (p[pl][3] << 16);
            *reinterpret_cast<uint2*>(sb + ((size_t)pl * ROWS + row) * LDB + 4 * c4) = o;
        }
    }
    __syncthreads();
    f32x16 acc[RT][4];
    for (int rt = 0; rt < RT; ++rt)
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 16; ++i) acc[rt][g][i] = 0.f;
    constexpr size_t PLANE = (size_t)KB16 * (NC / 32) * 64;
    const u32x4* wp = Wp + (size_t)(4 * w) * 64 + lane;
    u32x4 b[2][3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int g = 0; g < 4; ++g) b[0][pl][g] = wp[pl * PLANE + g * 64];
    // least significant products first
    constexpr int PA[9] = { 2, 1, 2, 0, 2, 1, 0, 1, 0 };
    constexpr int PB[9] = { 2, 2, 1, 2, 0, 1, 1, 0, 0 };
#pragma unroll 2
    for (int kb = 0; kb < KB16; ++kb) {
        if (kb + 1 < KB16) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int g = 0; g < 4; ++g) b[(kb + 1) & 1][pl][g] = wp[pl * PLANE + (size_t)(kb + 1) * (NC / 32) * 64 + g * 64];
        }
        u32x4 a[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                a[rt][pl] = *reinterpret_cast<const u32x4*>(sb + ((size_t)pl * ROWS + 32 * rt + li) * LDB + 16 * kb + 8 * lh);
#pragma unroll
        for (int t = 9 - NPROD; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[rt][PA[t]]),
                                                                         __builtin_bit_cast(bf16x8, b[kb & 1][PB[t]][g]),
                                                                         acc[rt][g], 0, 0, 0);
    }
    store_c<RT>(C, r0, store ? (int)((size_t)R - r0 < (size_t)ROWS ? (size_t)R - r0 : (size_t)ROWS) : -1, w, li, lh, acc);
}

// ---- the same with the A tile left in fp32 in LDS (66 KB: two workgroups per CU, as policy_step_kernel has it) and split
//      ON THE FLY: every wave reads 8 fp32 of its row per 16 k-steps and splits them in registers (4 x redundant across
//      the waves of the workgroup, no LDS beyond what the fp32 kernel uses) -----------------------------------------------
template <int ROWS, int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16_otf(const float* __restrict__ A, const u32x4* __restrict__ Wp,
                                                     float* __restrict__ C, int R, int store)
{
    constexpr int RT = ROWS / 32, LDA = K + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * (K / 4); i += 256) {
        const int row = i / (K / 4), c4 = i % (K / 4);
        f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (r0 + row < (size_t)R) v = *reinterpret_cast<const f32x4*>(A + (r0 + row) * K + 4 * c4);
        *reinterpret_cast<f32x4*>(sm + row * LDA + 4 * c4) = v;
    }
    __syncthreads();
    f32x16 acc[RT][4];
    for (int rt = 0; rt < RT; ++rt)
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 16; ++i) acc[rt][g][i] = 0.f;
    constexpr size_t PLANE = (size_t)KB16 * (NC / 32) * 64;
    const u32x4* wp = Wp + (size_t)(4 * w) * 64 + lane;
    u32x4 b[2][3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int g = 0; g < 4; ++g) b[0][pl][g] = wp[pl * PLANE + g * 64];
    constexpr int PA[9] = { 2, 1, 2, 0, 2, 1, 0, 1, 0 };
    constexpr int PB[9] = { 2, 2, 1, 2, 0, 1, 1, 0, 0 };
#pragma unroll 2
    for (int kb = 0; kb < KB16; ++kb) {
        if (kb + 1 < KB16) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int g = 0; g < 4; ++g) b[(kb + 1) & 1][pl][g] = wp[pl * PLANE + (size_t)(kb + 1) * (NC / 32) * 64 + g * 64];
        }
        u32x4 a[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float* src = sm + (32 * rt + li) * LDA + 16 * kb + 8 * lh;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
            unsigned p[3][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                split3(x0[q], p[0][q], p[1][q], p[2][q]);
                split3(x1[q], p[0][4 + q], p[1][4 + q], p[2][4 + q]);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int d = 0; d < 4; ++d) a[rt][pl][d] = p[pl][2 * d] | (p[pl][2 * d + 1] << 16);
        }
#pragma unroll
        for (int t = 9 - NPROD; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[rt][PA[t]]),
                                                                         __builtin_bit_cast(bf16x8, b[kb & 1][PB[t]][g]),
                                                                         acc[rt][g], 0, 0, 0);
    }
    store_c<RT>(C, r0, store ? (int)((size_t)R - r0 < (size_t)ROWS ? (size_t)R - r0 : (size_t)ROWS) : -1, w, li, lh, acc);
}

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                     \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 81920;
    std::vector<float> hA((size_t)R * K), hW((size_t)K * NC);
    unsigned s = 12345u;
    auto rnd = [&]() {   // values like the policy's: activations in [-1, 1], weights ~ +-0.1
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = 0.1f * rnd();
    float *dA, *dW, *dC;
    f32x4* dWq;
    u32x4* dWp;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dW, hW.size() * 4));
    CK(hipMalloc(&dC, (size_t)R * NC * 4));
    CK(hipMalloc(&dWq, (size_t)KB8 * (NC / 32) * 64 * 16));
    CK(hipMalloc(&dWp, (size_t)3 * KB16 * (NC / 32) * 64 * 16));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_f32, dim3((KB8 * (NC / 32) * 64 + 255) / 256), dim3(256), 0, 0, dW, dWq);
    hipLaunchKernelGGL(pack_bf16, dim3((KB16 * (NC / 32) * 64 + 255) / 256), dim3(256), 0, 0, dW, dWp);
    CK(hipDeviceSynchronize());
    // fp64 product of a row sample
    const int NS = 48;
    std::vector<int> rows(NS);
    for (int i = 0; i < NS; ++i) rows[i] = (int)(((long long)i * 1709 + 31) % R);
    rows[NS - 1] = R - 1;
    std::vector<double> ref((size_t)NS * NC, 0.0);
    for (int i = 0; i < NS; ++i)
        for (int k = 0; k < K; ++k) {
            const double a = hA[(size_t)rows[i] * K + k];
            for (int c = 0; c < NC; ++c) ref[(size_t)i * NC + c] += a * (double)hW[(size_t)k * NC + c];
        }
    std::vector<float> hC(NC);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flops = 2.0 * R * K * NC;
    auto report = [&](const char* name, float ms) -> int {
        double worst = 0, sum2 = 0, scale = 0;
        for (int i = 0; i < NS; ++i) {
            CK(hipMemcpy(hC.data(), dC + (size_t)rows[i] * NC, NC * 4, hipMemcpyDeviceToHost));
            for (int c = 0; c < NC; ++c) {
                const double d = fabs((double)hC[c] - ref[(size_t)i * NC + c]);
                worst = d > worst ? d : worst;
                sum2 += d * d;
                scale = fabs(ref[(size_t)i * NC + c]) > scale ? fabs(ref[(size_t)i * NC + c]) : scale;
            }
        }
        printf("%-44s %8.1f us  %6.1f TFLOP/s (fp32-equivalent)   max |err| %.3e  rms %.3e  (max |C| %.2f)\n", name, ms * 1e3,
               flops / (ms * 1e-3) / 1e12, worst, sqrt(sum2 / (NS * NC)), scale);
        return 0;
    };
#define RUN(name, kern, rows_per_wg, lds)                                                                           \
    do {                                                                                                            \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds))); \
        const int grid = (R + (rows_per_wg) - 1) / (rows_per_wg);                                                    \
        float best = 1e9f, best_ns = 1e9f;                                                                          \
        CK(hipMemset(dC, 0xff, (size_t)R * NC * 4));                                                                \
        for (int rep = 0; rep < 12; ++rep) {                                                                        \
            CK(hipEventRecord(e0));                                                                                 \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (lds), 0, dA, ARG, dC, R, rep < 6 ? 1 : 0);              \
            CK(hipEventRecord(e1));                                                                                 \
            CK(hipEventSynchronize(e1));                                                                            \
            float ms;                                                                                               \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                                   \
            if (rep && rep < 6 && ms < best) best = ms;                                                             \
            if (rep > 6 && ms < best_ns) best_ns = ms;                                                              \
        }                                                                                                           \
        CK(hipGetLastError());                                                                                      \
        if (report(name, best)) return 1;                                                                           \
        printf("%-44s %8.1f us  %6.1f TFLOP/s without the stores of C\n", "", best_ns * 1e3, flops / (best_ns * 1e-3) / 1e12); \
    } while (0)
    printf("C[%d x %d] = A[%d x %d] . W[%d x %d]; %.2f GFLOP\n", R, NC, R, K, K, NC, flops / 1e9);
#define ARG dWq
    RUN("fp32 MFMA 32x32x2, 64-row tiles (2 WG/CU)", (gemm_f32<64>), 64, 64 * (K + 4) * 4);
    RUN("fp32 MFMA 32x32x2, 32-row tiles", (gemm_f32<32>), 32, 32 * (K + 4) * 4);
#undef ARG
#define ARG dWp
    RUN("bf16x9 32x32x16, 64-row tiles (1 WG/CU)", (gemm_bf16<64, 9>), 64, 3 * 64 * (K + 8) * 2);
    RUN("bf16x9 32x32x16, 32-row tiles (3 WG/CU)", (gemm_bf16<32, 9>), 32, 3 * 32 * (K + 8) * 2);
    RUN("bf16x6 32x32x16, 64-row tiles (1 WG/CU)", (gemm_bf16<64, 6>), 64, 3 * 64 * (K + 8) * 2);
    RUN("bf16x6 32x32x16, 32-row tiles (3 WG/CU)", (gemm_bf16<32, 6>), 32, 3 * 32 * (K + 8) * 2);
    RUN("bf16x9, fp32 A tile split on the fly, 64 rows", (gemm_bf16_otf<64, 9>), 64, 64 * (K + 4) * 4);
    RUN("bf16x6, fp32 A tile split on the fly, 64 rows", (gemm_bf16_otf<64, 6>), 64, 64 * (K + 4) * 4);
#undef ARG
    return 0;
}
