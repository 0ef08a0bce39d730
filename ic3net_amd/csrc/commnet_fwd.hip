// commnet_fwd.hip — the NON-recurrent CommNet module (comm.py:127-129, 179-205, 220-224, 228-239) after the encoder, all
// communication passes in ONE launch — ic3_commnet_forward:
//     x = tanh(enc)                    enc = encoder(obs) incl. its bias (caller: sparse gather or dense GEMM)
//     h_0 = x
//     h_{i+1} = tanh(x + f_modules[i](h_i) + C_modules[i](comm(h_i)))      i = 0 .. comm_passes - 1
//     out = [log_softmax(heads_k(h)) ... | value_head(h)]
// A workgroup owns 64 rows = whole envs (the communication block mixes the agents of one env only).  The tile [comm | h]
// lives in LDS as the A operand; pass i's [C_i | F_i] (H x 2H, packed by ic3_commnet_pack like the C weights of
// policy_step.hip) streams from L2 as the B operand of v_mfma_f32_32x32x2_f32 — exact fp32 — x and the accumulators stay
// in registers in the MFMA C/D layout, tanh on the hardware transcendental unit (|error| <= ~2e-7, bar 1e-5).
// This is f3 coverage ("other policy variants"), not the headline path: one resident workgroup per CU at
// H >= 128, no store pacing, plain loops.
#include <hip/hip_runtime.h>

#include "ic3_common.hpp"

namespace ic3 {

typedef float cn_f32x4 __attribute__((ext_vector_type(4)));
typedef float cn_f32x16 __attribute__((ext_vector_type(16)));

struct CommnetArgs {
    const float* enc;          // [R][H]
    const float* wp;           // [passes][2H/8][H][2][4] packed [C_i | F_i]
    const float* bias;         // [passes][H]  C_i.bias + f_i.bias
    const float* head_w;       // [OT][H]
    const float* head_b;       // [OT]
    const int32_t* alive_in;   // [R] or null
    const int32_t* comm_in;    // [R] or null
    float* out;                // [R][OT]
    float* h_out;              // [R][H] or null: the final hidden state (tests)
    int E, N, EPT, passes, mode_avg, comm_zero, nheads, OT, a0, a1, a2, a3;
};

template <int H>
__global__ __launch_bounds__(2 * H, 1) void commnet_forward_kernel(const CommnetArgs a)
{
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, NT = 2 * H, H4 = H / 4, KB = K / 8, BM = 64;
    IC3_DYNAMIC_LDS(float, smem);
    float* const As = smem;                                      // [BM][LDA]: cols [0,H) comm, [H,2H) h
    cn_f32x4* const As4 = reinterpret_cast<cn_f32x4*>(smem);
    float* const sm = As + BM * LDA;                             // [BM] m_j = alive_j * comm_action_j
    float* const sscale = sm + BM;                               // [BM] per-env 1 / (n_alive - 1)
    int32_t* const sal = reinterpret_cast<int32_t*>(sscale + BM);   // [BM] alive flags
    float* const zl = reinterpret_cast<float*>(sal + BM);        // [BM][16] logits
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int col = 32 * w + li, N = a.N;
    const int e0 = blockIdx.x * a.EPT, nenv = min(a.EPT, a.E - e0), rows = nenv * N;
    const size_t r0 = (size_t)e0 * N;

    // ---- masks and per-env scale (comm.py:102-107,175-177,194-196) ---------------------------------------------------------
    if (tid < BM) {
        const bool in = tid < rows;
        const int al = (in && a.alive_in) ? a.alive_in[r0 + tid] : 1;
        const int cm = (in && a.comm_in) ? a.comm_in[r0 + tid] : 1;
        sm[tid] = in ? (float)(al * cm) : 0.f;
        sal[tid] = al;
    }
    // ---- x = tanh(enc) -> h half (comm.py:127-129) ---------------------------------------------------------------------------
    for (int idx = tid; idx < BM * H4; idx += NT) {
        const int row = idx / H4, c4 = idx - row * H4;
        cn_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (row < rows) {
            v = *reinterpret_cast<const cn_f32x4*>(a.enc + (r0 + row) * H + 4 * c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fast_tanh(v[q]);
        }
        As4[row * LDA4 + H4 + c4] = v;
    }
    __syncthreads();
    for (int el = tid; el < nenv; el += NT) {
        int n_alive = 0;
        for (int j = 0; j < N; ++j) n_alive += sal[el * N + j];
        sscale[el] = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
    }
    float xr[2][16];                                             // x in the MFMA C/D layout, kept for every pass
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) xr[rt][reg] = As[(32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * LDA + H + col];
    __syncthreads();

    for (int pass = 0; pass < a.passes; ++pass) {
        // ---- comm_j = m_j (S_e - m_j h_j) scale_e (closed form of comm.py:181-205) -> comm half -----------------------------
        {
            const int c4 = tid % H4;
            for (int el = tid / H4; el < nenv; el += NT / H4) {
                const cn_f32x4* hp = As4 + (el * N) * LDA4 + H4 + c4;
                const float scl = sscale[el];
                cn_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
                if (!a.comm_zero)
                    for (int i = 0; i < N; ++i) S += sm[el * N + i] * hp[i * LDA4];
                for (int j = 0; j < N; ++j) {
                    const float m = a.comm_zero ? 0.f : sm[el * N + j];
                    As4[(el * N + j) * LDA4 + c4] = m * (S - m * hp[j * LDA4]) * scl;
                }
            }
            for (int idx = rows * H4 + tid; idx < BM * H4; idx += NT) {
                const int row = idx / H4, c4p = idx - row * H4;
                As4[row * LDA4 + c4p] = cn_f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
        }
        __syncthreads();
        // ---- acc = [comm | h] . [C_i | F_i]^T : lane (li, lh) of wave w reads Wp[kb][32 w + li][lh] -> k = 8 kb + 4 lh + j --
        cn_f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][i] = 0.0f;
        const cn_f32x4* wp = reinterpret_cast<const cn_f32x4*>(a.wp) + (size_t)pass * (K * H / 4) + col * 2 + lh;
        constexpr int CH = 8;
        static_assert(KB % CH == 0, "2H / 8 is a multiple of 8");
        cn_f32x4 cb[2][CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) cb[0][k] = wp[(size_t)k * H * 2];
#pragma unroll 1
        for (int ch = 0; ch < KB / CH; ch += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = ch + half;
                if (c + 1 < KB / CH) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) cb[(half + 1) & 1][k] = wp[(size_t)((c + 1) * CH + k) * H * 2];
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int kb = c * CH + k;
                    const cn_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
                    const cn_f32x4 a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cb[half][k][j], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], cb[half][k][j], acc[1], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                         // every wave has read the old h
        // ---- h' = tanh(x + F h + C comm + biases) -> h half (comm.py:222-224) ------------------------------------------------
        const float b = a.bias[pass * H + col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                As[lr * LDA + H + col] = fast_tanh(xr[rt][reg] + acc[rt][reg] + b);
            }
        __syncthreads();
    }

    // ---- heads + value head (comm.py:228,239): weights -> rows [0, OT) of the comm half, logits -> zl ------------------------
    for (int i = tid; i < a.OT * H4; i += NT)
        As4[(i / H4) * LDA4 + i % H4] = reinterpret_cast<const cn_f32x4*>(a.head_w)[i];
    if (a.h_out) {
        for (int idx = tid; idx < rows * H4; idx += NT) {
            const int row = idx / H4, c4 = idx - row * H4;
            *reinterpret_cast<cn_f32x4*>(a.h_out + (r0 + row) * H + 4 * c4) = As4[row * LDA4 + H4 + c4];
        }
    }
    __syncthreads();
    for (int task = tid; task < rows * a.OT; task += NT) {
        const int row = task / a.OT, o = task - row * a.OT;
        const cn_f32x4* hp = As4 + row * LDA4 + H4;
        const cn_f32x4* wo = As4 + o * LDA4;
        cn_f32x4 s = { 0.f, 0.f, 0.f, 0.f };
        for (int k = 0; k < H4; ++k) s += hp[k] * wo[k];
        zl[row * 16 + o] = (s[0] + s[1]) + (s[2] + s[3]) + a.head_b[o];
    }
    __syncthreads();
    const int sizes[4] = { a.a0, a.a1, a.a2, a.a3 };
    for (int task = tid; task < rows * (a.nheads + 1); task += NT) {
        const int tr = task / (a.nheads + 1), hd = task - tr * (a.nheads + 1);
        const float* z = zl + tr * 16;
        float* orow = a.out + (r0 + tr) * a.OT;
        int off = 0;
        for (int i = 0; i < hd && i < a.nheads; ++i) off += sizes[i];
        if (hd == a.nheads) {
            orow[off] = z[off];
            continue;
        }
        const int A = sizes[hd];
        float mx = -INFINITY;
        for (int o = 0; o < A; ++o) mx = fmaxf(mx, z[off + o]);
        float sum = 0.0f;
        for (int o = 0; o < A; ++o) sum += __builtin_amdgcn_exp2f(1.4426950408889634f * (z[off + o] - mx));
        const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(sum);
        for (int o = 0; o < A; ++o) orow[off + o] = z[off + o] - lse;
    }
}

// Wp[kb][col][hh][j] = W[col][8 kb + 4 hh + j], W = [C | F] (H x 2H): the layout the kernel's B fragments read
__global__ void commnet_pack_kernel(const float* __restrict__ Cw, const float* __restrict__ Fw, float* __restrict__ Wp, int H)
{
    const long long n = (long long)H * 2 * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), hh = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int colx = (int)(rest % H), kb = (int)(rest / H);
        const int k = 8 * kb + 4 * hh + j;
        Wp[i] = k < H ? Cw[(size_t)colx * H + k] : Fw[(size_t)colx * H + (k - H)];
    }
}

}  // namespace ic3

extern "C" int ic3_commnet_forward_supported(int H, int N) { return (H == 64 || H == 128 || H == 256) && N >= 1 && N <= 64; }

extern "C" int ic3_commnet_pack(const float* C_weight, const float* f_weight, float* wp, int H, ic3_stream stream)
{
    using namespace ic3;
    if (!C_weight || !f_weight || !wp || H <= 0 || (H & 7)) return fail(-22, "ic3_commnet_pack: bad arguments");
    hipLaunchKernelGGL(commnet_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, C_weight, f_weight, wp, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_commnet_forward(const float* enc, int E, int N, int H, int comm_passes, const float* wp, const float* bias,
                                   const float* head_w, const float* head_b, const int32_t* head_sizes, int nheads,
                                   int mode_avg, int comm_zero, const int32_t* alive_in, const int32_t* comm_in, float* out,
                                   float* h_out, ic3_stream stream)
{
    using namespace ic3;
    if (!enc || !wp || !bias || !head_w || !head_b || !head_sizes || !out || E <= 0 || comm_passes < 1)
        return fail(-22, "ic3_commnet_forward: bad arguments");
    if (!ic3_commnet_forward_supported(H, N))
        return fail(-38, "ic3_commnet_forward: needs hid_size 64/128/256 and <= 64 agents per env");
    if (nheads < 1 || nheads > 4) return fail(-22, "ic3_commnet_forward: 1..4 action heads");
    CommnetArgs a{};
    a.enc = enc;
    a.wp = wp;
    a.bias = bias;
    a.head_w = head_w;
    a.head_b = head_b;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.h_out = h_out;
    a.E = E;
    a.N = N;
    a.EPT = 64 / N;
    a.passes = comm_passes;
    a.mode_avg = mode_avg;
    a.comm_zero = comm_zero;
    a.nheads = nheads;
    int sz[4] = { 0, 0, 0, 0 };
    a.OT = 1;
    for (int i = 0; i < nheads; ++i) {
        sz[i] = head_sizes[i];
        if (sz[i] < 1) return fail(-22, "ic3_commnet_forward: empty action head");
        a.OT += sz[i];
    }
    if (a.OT > 16) return fail(-22, "ic3_commnet_forward: more than 15 actions in total");
    a.a0 = sz[0];
    a.a1 = sz[1];
    a.a2 = sz[2];
    a.a3 = sz[3];
    const int tiles = (E + a.EPT - 1) / a.EPT;
    const size_t lds = ((size_t)64 * (2 * H + 4) + 3 * 64 + 64 * 16) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define IC3_CN(h)                                                                                                 \
    case h:                                                                                                       \
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h>), lds));               \
        hipLaunchKernelGGL(commnet_forward_kernel<h>, dim3(tiles), dim3(2 * h), lds, s, a);                       \
        break;
    switch (H) {
        IC3_CN(64)
        IC3_CN(128)
        IC3_CN(256)
    }
#undef IC3_CN
    IC3_HIP(hipGetLastError());
    return 0;
}
