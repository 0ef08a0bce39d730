// lstm_fused.hip — torch.nn.LSTMCell (comm.py:61,215) as ONE kernel for the rollout step (gfx950):
//     gates = [inp | h] · [W_ih | W_hh]^T + (b_ih + b_hh)        fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32)
//     c' = σ(f)·c + σ(i)·tanh(g) ;  h' = σ(o)·tanh(c')           in-register epilogue
// The (R x 4H) gate tensor is never materialised (the library-GEMM + pointwise pair wrote and re-read 4H floats
// per row).  MFMA-bound: 2·R·2H·4H flops at the 157 TFLOP/s fp32-matrix peak.
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

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int H>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 2 : 1) void lstm_fused_kernel(float* __restrict__ XH, int ldx,
                                                                               const f32x4* __restrict__ Wp,
                                                                               const float* __restrict__ bias,
                                                                               float* __restrict__ c, int R)
{
    constexpr int K = 2 * H, LDA = K + 1, BM = 64, NT = 2 * H;  // NT threads = 64 * (H/32) waves
    extern __shared__ __attribute__((aligned(16))) float As[];  // [BM][LDA]
    const int r0 = blockIdx.x * BM;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- stage the A tile: coalesced 16-byte global reads, scalar LDS writes (row stride breaks 16-B alignment) ----
    for (int idx = threadIdx.x; idx < BM * (K / 4); idx += NT) {
        const int row = idx / (K / 4), c4 = idx - row * (K / 4);
        f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (r0 + row < R) v = *reinterpret_cast<const f32x4*>(XH + (size_t)(r0 + row) * ldx + 4 * c4);
        float* dst = As + row * LDA + 4 * c4;
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
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
    f32x4 bq[4], bn[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = wp[(size_t)g * H * 2];
    const float* a0p = As + li * LDA + lh;
    const float* a1p = As + (32 + li) * LDA + lh;
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bn[g] = wp[(size_t)(kb + 1) * KB_STRIDE + (size_t)g * H * 2];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = a0p[8 * kb + 2 * j], a1 = a1p[8 * kb + 2 * j];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[g][j], acc[0][g], 0, 0, 0);
                acc[1][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[g][j], acc[1][g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = bn[g];
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    const int j = 32 * w + li;
    const float bi = bias[j], bf = bias[H + j], bg = bias[2 * H + j], bo = bias[3 * H + j];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (row < R) {
                const float gi = acc[rt][0][reg] + bi, gf = acc[rt][1][reg] + bf;
                const float gg = acc[rt][2][reg] + bg, go = acc[rt][3][reg] + bo;
                float* cp = c + (size_t)row * H + j;
                const float c1 = sigm(gf) * (*cp) + sigm(gi) * tanhf(gg);
                *cp = c1;
                XH[(size_t)row * ldx + H + j] = sigm(go) * tanhf(c1);
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
    if (!XH || !Wp || !bias || !c || R <= 0 || ldx < 2 * H || (ldx & 3))
        return ic3::fail(-22, "ic3_lstm_fused: bad arguments");
    const int blocks = (R + 63) / 64;
    const size_t lds = (size_t)64 * (2 * H + 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const ic3::f32x4* wp = reinterpret_cast<const ic3::f32x4*>(Wp);
    if (H == 128) {
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<128>, dim3(blocks), dim3(256), lds, s, XH, ldx, wp, bias, c, R);
    } else if (H == 64) {
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<64>, dim3(blocks), dim3(128), lds, s, XH, ldx, wp, bias, c, R);
    } else if (H == 256) {
        static bool attr_set = false;
        if (!attr_set) {
            IC3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ic3::lstm_fused_kernel<256>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(ic3::lstm_fused_kernel<256>, dim3(blocks), dim3(512), lds, s, XH, ldx, wp, bias, c, R);
    } else {
        return ic3::fail(-38, "ic3_lstm_fused: H must be 64, 128 or 256 (use ic3_lstm_cell after a library GEMM otherwise)");
    }
    IC3_HIP(hipGetLastError());
    return 0;
}
