// lstm_fused.hip — torch.nn.LSTMCell (comm.py:61,215) as ONE kernel for the rollout step (gfx950):
//     gates = [inp | h] · [W_ih | W_hh]^T + (b_ih + b_hh)        fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32)
//     c' = σ(f)·c + σ(i)·tanh(g) ;  h' = σ(o)·tanh(c')           in-register epilogue
// The (R x 4H) gate tensor is never materialised (the library-GEMM + pointwise pair wrote and re-read 4H floats
// per row).  MFMA-bound: 2·R·2H·4H flops at the 157 TFLOP/s fp32-matrix peak.
//
// STATUS (round 1, MI355X, R = 81920, H = 128; tools/microbench_lstm.py, profiles/r01/lstm_fused.txt): 223 us =
// 96 TFLOP/s for the whole cell standalone, vs 199 us hipBLASLt GEMM (108 TFLOP/s) + 44 us lstm_cell = 243 us;
// inside the rollout graph the library GEMM runs faster (178 us) and the step time is the same with either
// (0.60-0.61 ms), so the policy uses this kernel only when args.fused_lstm is set.  Without the phase skew below
// it took 240-245 us: the MFMA loop runs at the library's rate, but the A-tile staging and the epilogue (~45 us)
// of the two co-resident workgroups coincided instead of hiding under each other's MFMA loops.
//
// Decomposition (H = 128: 256 threads):
//   * one workgroup = 64 rows x ALL 4H gate columns: the workgroup is the only reader and the only writer of its
//     rows of the [inp | h] buffer XH, so h' can be written back in place (XH[:, H:]) once the A tile is in LDS;
//   * wave w owns hidden columns [32w, 32w+32) of all four gates -> 2 (row tiles) x 4 (gates) accumulators of
//     32x32 (128 registers); in the MFMA C layout a lane holds the SAME (row, column) in all four gate tiles, so
//     the LSTM nonlinearity needs no cross-lane traffic;
//   * A = XH tile (64 x 2H) staged once in LDS with row stride 2H+1 floats (A-fragment reads conflict-free:
//     bank = (row + k) mod 32); B = weights streamed from L2 in a pre-packed layout
//         Wp[k/8][col][k&1][(k>>1)&3]   (ic3_lstm_pack_weights)
//     so that one coalesced 16-byte load per lane feeds four MFMA k-steps; register double-buffered.
#include "ic3_common.hpp"

namespace ic3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int H>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 2 : 1) void lstm_fused_kernel(float* __restrict__ XH, int ldx,
                                                                               const f32x4* __restrict__ Wp,
                                                                               const float* __restrict__ bias,
                                                                               float* __restrict__ c, int R, int skew)
{
    constexpr int K = 2 * H, LDA = K + 1, BM = 64, NT = 2 * H;  // NT threads = 64 * (H/32) waves
    // Phase skew: the two workgroups resident on a CU would otherwise run their (memory-bound) staging/epilogue and
    // their (MFMA-bound) main loops at the same time.  Workgroups 256..511 — observed to be the second residency
    // slot of each CU in the first dispatch round; an assumption that only affects speed — start `skew` sleeps
    // late, and every later workgroup inherits the offset of the slot it replaces.
    if (skew > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(127);
    extern __shared__ __attribute__((aligned(16))) float As[];  // [BM][LDA]
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- stage the A tile: coalesced 16-byte global reads (all issued before the first LDS write), scalar LDS
    // writes (the 2H+1 row stride breaks 16-byte alignment).  Tiles are full except possibly the last one. ----
    const bool full = (r0 + BM <= R);   // workgroup-uniform
    {
        constexpr int PER = BM * (K / 4) / NT;   // float4s per thread (= 16)
        f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int row = idx / (K / 4), c4 = idx - row * (K / 4);
            if (full || r0 + row < R) v[i] = *reinterpret_cast<const f32x4*>(XH + (size_t)(r0 + row) * ldx + 4 * c4);
            else v[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int row = idx / (K / 4), c4 = idx - row * (K / 4);
            float* dst = As + row * LDA + 4 * c4;
            dst[0] = v[i].x;
            dst[1] = v[i].y;
            dst[2] = v[i].z;
            dst[3] = v[i].w;
        }
    }
    __syncthreads();

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][g][i] = 0.0f;

    // B fragments: lane (li, lh) of wave w reads Wp[kb][g*H + 32w + li][lh] -> four k-steps (k = 8kb + 2j + lh)
    const f32x4* wp = Wp + ((size_t)(32 * w + li) * 2 + lh);
    constexpr int KB = K / 8;
    constexpr size_t KB_STRIDE = (size_t)4 * H * 2;  // float4s per kb
    // Two weight-fragment buffers, refilled a full 32-MFMA block (2048 cycles) before they are needed: L2 latency
    // stays hidden behind the other buffer's MFMAs (no register copies, loop unrolled by two).
    f32x4 b0[4], b1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        b0[g] = wp[(size_t)g * H * 2];
        b1[g] = wp[KB_STRIDE + (size_t)g * H * 2];
    }
    const float* a0p = As + li * LDA + lh;
    const float* a1p = As + (32 + li) * LDA + lh;
    auto block = [&](const f32x4 (&bq)[4], int kb) {
        float a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] = a0p[8 * kb + 2 * j];
            a1[j] = a1p[8 * kb + 2 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bq[g][j], acc[0][g], 0, 0, 0);
                acc[1][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bq[g][j], acc[1][g], 0, 0, 0);
            }
        }
    };
    static_assert(KB % 2 == 0, "K/8 must be even");
    // sched_barrier(0) pins the phase order (the machine scheduler otherwise sinks the refill loads to just before
    // their first use, which exposes the full L2 latency every block).
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        block(b0, kb);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 2 < KB) {
#pragma unroll
            for (int g = 0; g < 4; ++g) b0[g] = wp[(size_t)(kb + 2) * KB_STRIDE + (size_t)g * H * 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        block(b1, kb + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 3 < KB) {
#pragma unroll
            for (int g = 0; g < 4; ++g) b1[g] = wp[(size_t)(kb + 3) * KB_STRIDE + (size_t)g * H * 2];
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  All 32 old cell values are
    // loaded up front (one memory latency, not 32); full tiles take a branch-free path. ----
    const int j = 32 * w + li;
    const float bi = bias[j], bf = bias[H + j], bg = bias[2 * H + j], bo = bias[3 * H + j];
    float cold[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            cold[rt][reg] = (full || row < R) ? c[(size_t)row * H + j] : 0.0f;
        }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const float gi = acc[rt][0][reg] + bi, gf = acc[rt][1][reg] + bf;
            const float gg = acc[rt][2][reg] + bg, go = acc[rt][3][reg] + bo;
            const float c1 = fast_sigmoid(gf) * cold[rt][reg] + fast_sigmoid(gi) * fast_tanh(gg);
            const float h1 = fast_sigmoid(go) * fast_tanh(c1);
            if (full || row < R) {
                c[(size_t)row * H + j] = c1;
                XH[(size_t)row * ldx + H + j] = h1;
            }
        }
    }
}

// Wcat = [W_ih | W_hh] is (4H x 2H) row-major; Wp[kb][col][h][j] = Wcat[col][8 kb + 2 j + h]
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, float* __restrict__ Wp,
                                 int H)
{
    const int K = 2 * H;
    const long long n = (long long)4 * H * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(i & 3), h = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int col = (int)(rest % (4 * H)), kb = (int)(rest / (4 * H));
        const int k = 8 * kb + 2 * jj + h;
        Wp[i] = k < H ? w_ih[(size_t)col * H + k] : w_hh[(size_t)col * H + (k - H)];
    }
}

}  // namespace ic3

extern "C" int ic3_lstm_pack_weights(const float* w_ih, const float* w_hh, float* Wp, int H, ic3_stream stream)
{
    if (!w_ih || !w_hh || !Wp || H <= 0 || (H & 31)) return ic3::fail(-22, "ic3_lstm_pack_weights: H must be a multiple of 32");
    hipLaunchKernelGGL(ic3::lstm_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, Wp, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_lstm_fused(float* XH, int ldx, const float* Wp, const float* bias, float* c, int R, int H,
                              ic3_stream stream)
{
    const int skew = 6;   // x ~3.4 us (s_sleep 127) ~ 2/3 of a tile; measured 0/2/4/6/8 -> 240/241/229/223/225 us
    if (!XH || !Wp || !bias || !c || R <= 0 || ldx < 2 * H || (ldx & 3))
        return ic3::fail(-22, "ic3_lstm_fused: bad arguments");
    const int blocks = (R + 63) / 64;
    const size_t lds = (size_t)64 * (2 * H + 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const ic3::f32x4* wp = reinterpret_cast<const ic3::f32x4*>(Wp);
    if (H == 128) {
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<128>, dim3(blocks), dim3(256), lds, s, XH, ldx, wp, bias, c, R, skew);
    } else if (H == 64) {
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<64>, dim3(blocks), dim3(128), lds, s, XH, ldx, wp, bias, c, R, skew);
    } else if (H == 256) {
        static bool attr_set = false;
        if (!attr_set) {
            IC3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ic3::lstm_fused_kernel<256>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<256>, dim3(blocks), dim3(512), lds, s, XH, ldx, wp, bias, c, R, skew);
    } else {
        return ic3::fail(-38, "ic3_lstm_fused: H must be 64, 128 or 256 (use ic3_lstm_cell after a library GEMM otherwise)");
    }
    IC3_HIP(hipGetLastError());
    return 0;
}
