// policy_pack.hip — the weight layouts ic3_policy_step streams (ic3_policy_pack: k-major float4 over the four gates;
// ic3_policy_pack_split / _split_bwd: three exact bf16 planes in MFMA fragment order, forward and backward product) and
// ic3_gate_product_probe, the gate product of comm.py:215's LSTMCell ALONE through the operand layouts and instruction order of
// policy_step_kernel's gate loops (tests/test_gate_split_gpu.py: the arithmetic ruling of DESIGN.md section 0 at the edges).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ic3_common.hpp"
#include "ps_common.hpp"

namespace ic3 {

// ---- ic3_gate_product_probe: the gate product ALONE (pre-activations without the bias), through the operand layouts, the
// activation split and the per-accumulator order of matrix instructions of policy_step_kernel's two gate loops — (k ascending;
// fp32: one v_mfma_f32_32x32x2_f32 per k; split: per 16-k block weight plane outer, then gate, then the activation terms
// least significant first) — without their prefetch rings and store slots.  What the arithmetic of the two modes IS can be
// measured with it on operands no rollout produces (edge magnitudes, tests/test_gate_split_gpu.py).
template <int H, int SPLIT>
__global__ __launch_bounds__(2 * H) void gate_product_probe_kernel(const float* __restrict__ xh, const ps_f32x4* l_wp, const void* l_wp3,
                                                                   float* __restrict__ gates, int R)
{
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, BM = 64, NT = 2 * H, NW = H / 32, KB = K / 8, KB16 = K / 16;
    IC3_DYNAMIC_LDS(float, smem);
    ps_f32x4* const As4 = reinterpret_cast<ps_f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5, col = 32 * w + li;
    const size_t r0 = (size_t)blockIdx.x * BM;
    for (int idx = tid; idx < BM * (K / 4); idx += NT) {
        const int row = idx / (K / 4), c4 = idx - row * (K / 4);
        ps_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (r0 + row < (size_t)R) v = *reinterpret_cast<const ps_f32x4*>(xh + (r0 + row) * K + 4 * c4);
        As4[row * LDA4 + c4] = v;
    }
    __syncthreads();
    ps_f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][gt][i] = 0.0f;
    if constexpr (SPLIT != 0) {
        const __amdgpu_buffer_rsrc_t rg3 = make_rsrc(l_wp3, (uint32_t)((size_t)3 * K * 4 * H * 2));
        const int g3lane = (w * 64 + lane) * 16;
        constexpr int GSTRIDE = NW * 64 * 16;
#pragma unroll 1
        for (int kb = 0; kb < KB16; ++kb) {
            ps_u32x4 ap[2][3];
            const ps_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh;
            ps_split_frag(s0[0], s0[1], ap[0]);
            const ps_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh;
            ps_split_frag(s1[0], s1[1], ap[1]);
            // (product order of block3: weight plane group 0 pass by pass, most significant activation plane first; groups
            //  1 and 2 fragment by fragment, least significant activation plane first)
            ps_u32x4 bq0[4];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
                bq0[gt] = __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, (kb * 4 + gt) * GSTRIDE, 0);
#pragma unroll
            for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(ps_bf16x8, ap[rt][pa]), __builtin_bit_cast(ps_bf16x8, bq0[gt]), acc[rt][gt], 0, 0, 0);
#pragma unroll
            for (int pb = 1; pb < 3; ++pb)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {
                    const ps_u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, ((pb * KB16 + kb) * 4 + gt) * GSTRIDE, 0);
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
                            acc[rt][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(ps_bf16x8, ap[rt][pa]), __builtin_bit_cast(ps_bf16x8, bq), acc[rt][gt], 0, 0, 0);
                }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rgw = make_rsrc(l_wp, (uint32_t)((size_t)K * 4 * H * sizeof(float)));
        const int glane = (4 * lh * H + col) * 16;
#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            const ps_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh], a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const ps_f32x4 wk = buf_load_b128(rgw, glane, (8 * kb + j) * (H * 16));
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {
                    acc[0][gt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], wk[gt], acc[0][gt], 0, 0, 0);
                    acc[1][gt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], wk[gt], acc[1][gt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const size_t row = r0 + 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (row < (size_t)R) gates[row * 4 * H + (size_t)gt * H + col] = acc[rt][gt][reg];
            }
}

// Wp[kb][col][hh][j] = W[col][8 kb + 4 hh + j], W = [Wa | Wb] (C x (Ka + Kb)) row-major halves
__global__ void policy_pack_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb, float* __restrict__ Wp,
                                   int C, int Ka, int Kb)
{
    const int Kt = Ka + Kb;
    const long long n = (long long)C * Kt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), hh = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int colx = (int)(rest % C), kb = (int)(rest / C);
        const int k = 8 * kb + 4 * hh + j;
        Wp[i] = k < Ka ? Wa[(size_t)colx * Ka + k] : Wb[(size_t)colx * Kb + (k - Ka)];
    }
}

// Wq[k][c] = float4 (W[c][k], W[H + c][k], W[2H + c][k], W[3H + c][k]), W = [w_ih | w_hh] (4H x 2H): the gate GEMM's B
// operand, one k-step of all four gates of a hidden column per 16-byte load
__global__ void policy_pack_gates_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, float* __restrict__ Wq, int H)
{
    const long long n = (long long)8 * H * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int gt = (int)(i & 3);
        const long long rest = i >> 2;
        const int c = (int)(rest % H), k = (int)(rest / H);
        const size_t row = (size_t)gt * H + c;
        Wq[i] = k < H ? w_ih[row * H + k] : w_hh[row * H + (k - H)];
    }
}

// gate_split: Wp[plane][kb16][gate][wave][lane] = 8 x bf16 { W_plane[gate * H + 32 wave + li][16 kb16 + 8 lh + i] },
// W = [w_ih | w_hh] (4H x 2H), the three planes an exact split of every weight
__global__ void policy_pack_split_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                         ps_u32x4* __restrict__ Wp, int H)
{
    const int NWv = H / 32, KB16 = 2 * H / 16;
    const long long per = (long long)KB16 * 4 * NWv * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long long rest = i >> 6;
        const int wv = (int)(rest % NWv);
        rest /= NWv;
        const int gt = (int)(rest & 3), kb = (int)(rest >> 2);
        const int li = lane & 31, lh = lane >> 5;
        const size_t row = (size_t)gt * H + 32 * wv + li;
        unsigned p[3][8];
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * kb + 8 * lh + q;
            ps_split3(k < H ? w_ih[row * H + k] : w_hh[row * H + (k - H)], p[0][q], p[1][q], p[2][q]);
        }
        for (int pl = 0; pl < 3; ++pl) {
            ps_u32x4 v;
            for (int d = 0; d < 4; ++d) v[d] = p[pl][2 * d] | (p[pl][2 * d + 1] << 16);
            Wp[(size_t)pl * per + i] = v;
        }
    }
}

// The same planes for the BACKWARD of the gate product (ic3_lstm_gates_backward_dx: [d inp | d h] = dgates . [W_ih | W_hh]):
// Wb[plane][kb16][ct][lane] = 8 x bf16 { W_plane[16 kb16 + 8 lh + i][32 ct + li] }, W = [w_ih | w_hh] (4H x 2H) — k runs over
// the 4H gate rows, the output column over the 2H inputs
__global__ void policy_pack_split_bwd_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                             ps_u32x4* __restrict__ Wp, int H)
{
    const int NCT = 2 * H / 32, KB16B = 4 * H / 16;
    const long long per = (long long)KB16B * NCT * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long rest = i >> 6;
        const int ct = (int)(rest % NCT), kb = (int)(rest / NCT);
        const int li = lane & 31, lh = lane >> 5;
        const int n = 32 * ct + li;
        unsigned p[3][8];
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * kb + 8 * lh + q;
            ps_split3(n < H ? w_ih[(size_t)k * H + n] : w_hh[(size_t)k * H + (n - H)], p[0][q], p[1][q], p[2][q]);
        }
        for (int pl = 0; pl < 3; ++pl) {
            ps_u32x4 v;
            for (int d = 0; d < 4; ++d) v[d] = p[pl][2 * d] | (p[pl][2 * d + 1] << 16);
            Wp[(size_t)pl * per + i] = v;
        }
    }
}
}  // namespace ic3

using namespace ic3;

extern "C" int ic3_policy_pack(const float* c_weight, const float* w_ih, const float* w_hh, float* c_wp, float* lstm_wp,
                               int H, ic3_stream stream)
{
    if (!c_weight || !w_ih || !w_hh || !c_wp || !lstm_wp || H <= 0 || (H & 31))
        return fail(-22, "ic3_policy_pack: H must be a positive multiple of 32");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(policy_pack_kernel, dim3(64), dim3(256), 0, s, c_weight, (const float*)nullptr, c_wp, H, H, 0);
    hipLaunchKernelGGL(policy_pack_gates_kernel, dim3(256), dim3(256), 0, s, w_ih, w_hh, lstm_wp, H);
    IC3_HIP(hipGetLastError());
    return 0;
}


extern "C" int ic3_policy_pack_split(const float* w_ih, const float* w_hh, void* lstm_wp3, int H, ic3_stream stream)
{
    if (!w_ih || !w_hh || !lstm_wp3 || H <= 0 || (H % 32))
        return fail(-22, "ic3_policy_pack_split: H must be a positive multiple of 32");
    hipLaunchKernelGGL(policy_pack_split_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh,
                       reinterpret_cast<ps_u32x4*>(lstm_wp3), H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_policy_pack_split_bwd(const float* w_ih, const float* w_hh, void* lstm_wp3_bwd, int H, ic3_stream stream)
{
    if (!w_ih || !w_hh || !lstm_wp3_bwd || H <= 0 || (H % 32))
        return fail(-22, "ic3_policy_pack_split_bwd: H must be a positive multiple of 32");
    hipLaunchKernelGGL(policy_pack_split_bwd_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh,
                       reinterpret_cast<ps_u32x4*>(lstm_wp3_bwd), H);
    IC3_HIP(hipGetLastError());
    return 0;
}

template <int H>
static int launch_gate_probe(const float* xh, const float* lstm_wp, const void* lstm_wp3, float* gates, int R, hipStream_t s)
{
    const size_t lds = (size_t)64 * (2 * H + 4) * sizeof(float);
    const dim3 grid((R + 63) / 64), block(2 * H);
    if (lstm_wp3) {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(&gate_product_probe_kernel<H, 1>), lds));
        hipLaunchKernelGGL((gate_product_probe_kernel<H, 1>), grid, block, lds, s, xh, (const ps_f32x4*)nullptr, lstm_wp3, gates, R);
    } else {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(&gate_product_probe_kernel<H, 0>), lds));
        hipLaunchKernelGGL((gate_product_probe_kernel<H, 0>), grid, block, lds, s, xh, reinterpret_cast<const ps_f32x4*>(lstm_wp),
                           (const void*)nullptr, gates, R);
    }
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_gate_product_probe(const float* xh, const float* lstm_wp, const void* lstm_wp3, float* gates, int R, int H,
                                      ic3_stream stream)
{
    if (!xh || !gates || (!lstm_wp && !lstm_wp3) || R <= 0) return fail(-22, "ic3_gate_product_probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) return launch_gate_probe<128>(xh, lstm_wp, lstm_wp3, gates, R, s);
    if (H == 64) return launch_gate_probe<64>(xh, lstm_wp, lstm_wp3, gates, R, s);
    if (H == 256) return launch_gate_probe<256>(xh, lstm_wp, lstm_wp3, gates, R, s);
    return fail(-38, "ic3_gate_product_probe: hid_size 64 / 128 / 256");
}
