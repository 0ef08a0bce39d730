// comm_fused.hip — the communication block of CommNetMLP (comm.py:181-206) as ONE kernel on the rollout path (gfx950):
//     comm_j = m_j (S_e - m_j h_j) * scale_e            closed form of the N x N x H mask chain (policy_ops.hip)
//     inp_j += comm_j · C.weight^T                        fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 products)
// in place on the [inp | h] buffer XH.  Replaces comm_masked_mean_kernel (writes R*H floats) + a library GEMM that
// re-reads them: the comm rows only ever exist in LDS.  Traffic: read h (R*H*4), read-modify-write inp (2*R*H*4).
//
// Decomposition: one workgroup = a tile of whole envs, EPT = 64/N of them (<= 64 rows, padded with zero rows); the
// comm rows are built in registers (H/4 lanes per env: masked sum, then the rows re-read from L1/L2) and written to LDS
// (row stride H+1: conflict-free A-fragment reads); wave w owns output columns [32w, 32w+32) of both
// 32-row tiles.  All B (weight) fragments of a wave — K/8 float4 per lane — and the old inp values of its C fragment
// are requested right after the staging phase and before the MFMA loop.  Weights are pre-packed like
// ic3_lstm_pack_weights:  Wp[k/8][col][k&1][(k>>1)&3] = C.weight[col][k]
//
// STATUS (round 1, MI355X, E = 8192, N = 10, H = 128; tools/microbench_comm.py): 49 us standalone vs 62 us for
// comm_masked_mean + the library GEMM, but the same step time inside the rollout graph (0.559 ms either way: there h
// comes from HBM, not the Infinity Cache, and the per-workgroup phases staging 22 us / MFMA 22 us / store 7 us — measured
// by disabling them one at a time — do not overlap across the 1.8 rounds of workgroups).  Hence opt-in
// (args.fused_comm); the default stays comm_masked_mean + hipBLASLt/rocBLAS.
#include "ic3_common.hpp"

namespace ic3 {

typedef float cf_f32x4 __attribute__((ext_vector_type(4)));
typedef float cf_f32x16 __attribute__((ext_vector_type(16)));

template <int H, int BM>
__global__ __launch_bounds__(2 * H) void comm_c_kernel(float* __restrict__ XH, int ldx, const cf_f32x4* __restrict__ Wp,
                                                       const int32_t* __restrict__ alive,
                                                       const int32_t* __restrict__ comm_action, int E, int N, int EPT,
                                                       int mode_avg)
{
    constexpr int K = H, LDA = K + 1, NT = 2 * H, KB = K / 8, RT = BM / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [BM][LDA]  comm rows
    float* sm = As + BM * LDA;              // [BM] m_j
    float* sscale = sm + BM;                // [EPT]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int e0 = blockIdx.x * EPT;
    const int nenv = min(EPT, E - e0);
    const int rows = nenv * N;              // valid rows of this tile
    const size_t r0 = (size_t)e0 * N;

    // masks and per-env scale (comm.py:102-107,194-196; quirks Q21/Q23 as in comm_masked_mean_kernel)
    for (int r = threadIdx.x; r < BM; r += NT) {
        float m = 0.f;
        if (r < rows) m = (float)((alive ? alive[r0 + r] : 1) * (comm_action ? comm_action[r0 + r] : 1));
        sm[r] = m;
    }
    for (int el = threadIdx.x; el < nenv; el += NT) {
        int n_alive = 0;
        for (int j = 0; j < N; ++j) n_alive += alive ? alive[r0 + (size_t)el * N + j] : 1;
        sscale[el] = (mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
    }
    __syncthreads();
    // comm rows straight from global memory into LDS: H/4 lanes per env, each owning 4 hidden columns — pass 1 the
    // masked sum S_e (registers), pass 2 (rows re-read from L1/L2) comm_j = m_j (S_e - m_j h_j) scale_e
    {
        constexpr int H4 = H / 4;
        const int c4 = threadIdx.x % H4;
        for (int el = threadIdx.x / H4; el < nenv; el += NT / H4) {
            const float* hp = XH + (r0 + (size_t)el * N) * ldx + H + 4 * c4;
            const float sc = sscale[el];
            if (N <= 16) {
                // all rows of the env in flight at once (HBM latency paid once), kept in registers for pass 2
                cf_f32x4 hv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    hv[i] = i < N ? *reinterpret_cast<const cf_f32x4*>(hp + (size_t)i * ldx) : cf_f32x4{ 0.f, 0.f, 0.f, 0.f };
                cf_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i < N) S += sm[el * N + i] * hv[i];
#pragma unroll
                for (int jrow = 0; jrow < 16; ++jrow) {
                    if (jrow < N) {
                        const float m = sm[el * N + jrow];
                        const cf_f32x4 v = m * (S - m * hv[jrow]) * sc;
                        float* dst = As + (el * N + jrow) * LDA + 4 * c4;
                        dst[0] = v.x;
                        dst[1] = v.y;
                        dst[2] = v.z;
                        dst[3] = v.w;
                    }
                }
                continue;
            }
            cf_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
            for (int i = 0; i < N; ++i) S += sm[el * N + i] * *reinterpret_cast<const cf_f32x4*>(hp + (size_t)i * ldx);
            for (int jrow = 0; jrow < N; ++jrow) {
                const float m = sm[el * N + jrow];
                const cf_f32x4 v = m * (S - m * *reinterpret_cast<const cf_f32x4*>(hp + (size_t)jrow * ldx)) * sc;
                float* dst = As + (el * N + jrow) * LDA + 4 * c4;
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
        }
        for (int idx = rows * H + threadIdx.x; idx < BM * H; idx += NT) As[(idx / H) * LDA + idx % H] = 0.f;   // pad rows
    }
    // B fragments: lane (li, lh) of wave w reads Wp[kb][32w + li][lh] -> k = 8kb + 2j + lh, j = 0..3
    cf_f32x4 b[KB];
    {
        const cf_f32x4* wp = Wp + ((size_t)(32 * w + li) * 2 + lh);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb] = wp[(size_t)kb * H * 2];
    }

    __syncthreads();
    // the old inp values of this lane's C fragment: issued now, consumed after the MFMA loop
    const int col = 32 * w + li;
    float old[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            old[rt][reg] = lr < rows ? XH[(r0 + lr) * ldx + col] : 0.0f;
        }

    cf_f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rt][i] = 0.0f;
    const float* a0p = As + li * LDA + lh;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[rt * 32 * LDA + 8 * kb + 2 * j], b[kb][j], acc[rt], 0, 0, 0);
        }
    }

    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); inp = old + acc
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (lr < rows) XH[(r0 + lr) * ldx + col] = old[rt][reg] + acc[rt][reg];
        }
}

// Wp[kb][col][h][j] = Cw[col][8 kb + 2 j + h]     (Cw = C.weight, H x H row-major)
__global__ void comm_pack_kernel(const float* __restrict__ Cw, float* __restrict__ Wp, int H)
{
    const long long n = (long long)H * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(i & 3), h = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int col = (int)(rest % H), kb = (int)(rest / H);
        Wp[i] = Cw[(size_t)col * H + 8 * kb + 2 * jj + h];
    }
}

}  // namespace ic3

extern "C" int ic3_comm_pack_weights(const float* Cw, float* Wp, int H, ic3_stream stream)
{
    if (!Cw || !Wp || H <= 0 || (H & 31)) return ic3::fail(-22, "ic3_comm_pack_weights: H must be a multiple of 32");
    hipLaunchKernelGGL(ic3::comm_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, Cw, Wp, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_comm_fused(float* XH, int ldx, const float* Wp, const int32_t* alive, const int32_t* comm_action, int E,
                              int N, int H, int mode_avg, ic3_stream stream)
{
    if (!XH || !Wp || E <= 0 || N <= 0 || ldx < 2 * H || (ldx & 3)) return ic3::fail(-22, "ic3_comm_fused: bad arguments");
    if (N > 64) return ic3::fail(-38, "ic3_comm_fused: more than 64 agents per env (use ic3_comm_masked_mean + a GEMM)");
    const int BMr = 64;
    const int EPT = BMr / N, tiles = (E + EPT - 1) / EPT;
    const size_t lds = ((size_t)BMr * (H + 1) + BMr + EPT) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const ic3::cf_f32x4* wp = reinterpret_cast<const ic3::cf_f32x4*>(Wp);
    if (H == 128) {
        hipLaunchKernelGGL((ic3::comm_c_kernel<128, 64>), dim3(tiles), dim3(256), lds, s, XH, ldx, wp, alive, comm_action, E, N,
                           EPT, mode_avg);
    } else if (H == 64) {
        hipLaunchKernelGGL((ic3::comm_c_kernel<64, 64>), dim3(tiles), dim3(128), lds, s, XH, ldx, wp, alive, comm_action, E, N,
                           EPT, mode_avg);
    } else if (H == 256) {
        static bool attr_set = false;
        if (!attr_set) {
            IC3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ic3::comm_c_kernel<256, 64>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL((ic3::comm_c_kernel<256, 64>), dim3(tiles), dim3(512), lds, s, XH, ldx, wp, alive, comm_action, E, N,
                           EPT, mode_avg);
    } else {
        return ic3::fail(-38, "ic3_comm_fused: H must be 64, 128 or 256");
    }
    IC3_HIP(hipGetLastError());
    return 0;
}
