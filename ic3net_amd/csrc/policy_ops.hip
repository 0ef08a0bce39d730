// policy_ops.hip — the two custom policy-side ops of the rollout step (gfx950):
//   ic3_comm_masked_mean : CommNetMLP communication block, /root/reference/comm.py:181-205
//   ic3_sample_actions   : select_action, /root/reference/action_utils.py:32-36
// plus synthetic uniform actions for env-only benchmarks.
#include <type_traits>

#include "env_device.hpp"
#include <algorithm>

#include "ic3_common.hpp"

namespace ic3 {

// comm.py:181-205 builds an (B,N,N,H) expand of h, multiplies by (1-eye), 1/(n_alive-1), the sender
// mask and the receiver mask, and sums over senders: O(N^2 H) memory traffic in fp64.  Closed form
// per env (SURVEY B.5 i, probed exact to 2e-16 against the reference):
//     m_j  = alive_j * comm_action_j
//     out_j = m_j * (S - m_j * h_j) * scale,   S = sum_i m_i h_i,   scale = 1/(n_alive-1) if avg && n_alive>1
// One workgroup per env, one lane per hidden column: h rows are read coalesced (and re-read from L1/L2
// for the second pass), 2*N*H*4 algorithmic bytes per env.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// H4 = H/4 lanes per env (each lane owns 4 hidden columns as one 16-byte access); a 256-thread workgroup holds
// 256/H4 envs.  h rows may be strided (ldh floats) so the op can read the [inp | h] buffer of the fused LSTM path.
template <int NMAX>
__global__ __launch_bounds__(256) void comm_masked_mean_kernel(const float* __restrict__ h, int ldh,
                                                               const int32_t* __restrict__ alive,
                                                               const int32_t* __restrict__ comm_action,
                                                               float* __restrict__ out, int E, int N, int H4,
                                                               int mode_avg, int mask_self,
                                                               const float* __restrict__ addend = nullptr, int lda = 0,
                                                               const float* __restrict__ out_scale = nullptr)
{
    const int per_block = blockDim.x / H4;
    const int e = blockIdx.x * per_block + threadIdx.x / H4;
    const int k = threadIdx.x % H4;
    if (e >= E || threadIdx.x >= per_block * H4) return;
    int n_alive = 0;
    for (int j = 0; j < N; ++j) n_alive += alive ? alive[(size_t)e * N + j] : 1;   // comm.py:102-107, quirk Q21
    const float scale = (mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;  // comm.py:194-196, Q23
    f32x4* oe = reinterpret_cast<f32x4*>(out + (size_t)e * N * H4 * 4) + k;
    // ic3_comm_masked_mean_add: out = addend + the block's output (addend rows may be a strided column slice)
    auto add = [&](int j) {
        return addend ? *reinterpret_cast<const f32x4*>(addend + ((size_t)e * N + j) * lda + 4 * k) : f32x4{ 0.f, 0.f, 0.f, 0.f };
    };
    // out_scale [E*N] (with addend only): every output row times its factor — the update half's collection mode drops the
    // gradient that would cross an episode boundary right where it is produced
    auto sc = [&](int j) { return out_scale ? out_scale[(size_t)e * N + j] : 1.0f; };
    if (!mask_self) {  // comm_mask_zero: comm.py:40-41 -> all-zero communication
        for (int j = 0; j < N; ++j) oe[(size_t)j * H4] = out_scale ? add(j) * sc(j) : add(j);
        return;
    }
    f32x4 S = { 0.f, 0.f, 0.f, 0.f };
    if (NMAX > 0 && N <= NMAX) {
        // all rows of the env requested at once and kept in registers: one HBM read of h (the two-pass loop below
        // re-fetched every row — PMC FETCH_SIZE showed 2x the algorithmic bytes — and serialised the latencies)
        constexpr int NR = NMAX > 0 ? NMAX : 1;
        f32x4 hv[NR];
        float mm[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i < N) {
                hv[i] = *reinterpret_cast<const f32x4*>(h + ((size_t)e * N + i) * ldh + 4 * k);
                mm[i] = (float)((alive ? alive[(size_t)e * N + i] : 1) * (comm_action ? comm_action[(size_t)e * N + i] : 1));
            } else {
                hv[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
                mm[i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (i < N) S = mask_fma4(mm[i], hv[i], S);            // (one component per instruction: ic3_common.hpp)
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (j < N) {
                const f32x4 v = addend ? add(j) + comm_out4(mm[j], S, hv[j], scale) : comm_out4(mm[j], S, hv[j], scale);
                oe[(size_t)j * H4] = out_scale ? v * sc(j) : v;
            }
        return;
    }
    for (int i = 0; i < N; ++i) {
        const int m = (alive ? alive[(size_t)e * N + i] : 1) * (comm_action ? comm_action[(size_t)e * N + i] : 1);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + ((size_t)e * N + i) * ldh + 4 * k);
        S = mask_fma4((float)m, hv, S);
    }
    for (int j = 0; j < N; ++j) {
        const float m = (float)((alive ? alive[(size_t)e * N + j] : 1) *
                                (comm_action ? comm_action[(size_t)e * N + j] : 1));
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + ((size_t)e * N + j) * ldh + 4 * k);
        const f32x4 v = addend ? add(j) + comm_out4(m, S, hv, scale) : comm_out4(m, S, hv, scale);
        oe[(size_t)j * H4] = out_scale ? v * sc(j) : v;
    }
}

// scalar fallback for H % 4 != 0
__global__ __launch_bounds__(256) void comm_masked_mean_scalar_kernel(const float* __restrict__ h, int ldh,
                                                                      const int32_t* __restrict__ alive,
                                                                      const int32_t* __restrict__ comm_action,
                                                                      float* __restrict__ out, int N, int H,
                                                                      int mode_avg, int mask_self)
{
    const int e = blockIdx.x;
    float* oe = out + (size_t)e * N * H;
    int n_alive = 0;
    for (int j = 0; j < N; ++j) n_alive += alive ? alive[(size_t)e * N + j] : 1;
    const float scale = (mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        float S = 0.0f;
        for (int i = 0; i < N; ++i) {
            const int m = (alive ? alive[(size_t)e * N + i] : 1) * (comm_action ? comm_action[(size_t)e * N + i] : 1);
            S += (float)m * h[((size_t)e * N + i) * ldh + k];
        }
        for (int j = 0; j < N; ++j) {
            const float m = (float)((alive ? alive[(size_t)e * N + j] : 1) *
                                    (comm_action ? comm_action[(size_t)e * N + j] : 1));
            oe[(size_t)j * H + k] = mask_self ? m * (S - m * h[((size_t)e * N + j) * ldh + k]) * scale : 0.0f;
        }
    }
}

// torch.nn.LSTMCell pointwise half (comm.py:215, gate order i,f,g,o): gates [R][4H] already hold
// W_ih x + b_ih + W_hh h + b_hh.  c is updated in place, h' is written with row stride ldh (into the
// [inp | h] buffer).  HBM-bound: reads 4H+H, writes 2H floats per row.

__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, float* __restrict__ c,
                                                        float* __restrict__ h_out, int ldh, int R, int H4)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * H4) return;
    const int row = (int)(idx / H4), k = (int)(idx - (long long)row * H4);
    const f32x4* g = reinterpret_cast<const f32x4*>(gates + (size_t)row * 16 * H4);
    const f32x4 gi = g[k], gf = g[H4 + k], gg = g[2 * H4 + k], go = g[3 * H4 + k];
    f32x4* cp = reinterpret_cast<f32x4*>(c + (size_t)row * 4 * H4) + k;
    const f32x4 c0 = *cp;
    f32x4 c1, h1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c1[q] = fast_sigmoid(gf[q]) * c0[q] + fast_sigmoid(gi[q]) * fast_tanh(gg[q]);
        h1[q] = fast_sigmoid(go[q]) * fast_tanh(c1[q]);
    }
    *cp = c1;
    *reinterpret_cast<f32x4*>(h_out + (size_t)row * ldh + 4 * k) = h1;
}

// Backward of the pointwise half of LSTMCell for the update half (trainer.py:128-225 through comm.py:215): from the
// recomputed gate pre-activations, c_{t-1}, dL/dh_t (heads + what step t+1 sent back) and dL/dc_t -> dL/dgates (the
// operand of the two weight-gradient / input-gradient GEMMs) and dL/dc_{t-1}.  Nothing of the forward is kept for it:
// i, f, g, o, c_t and tanh(c_t) are re-evaluated here (the same hardware exp / rcp forms as the forward kernels).
// HBM-bound: reads 4H + 3H, writes 4H + H floats per row.
// `dbias` (optional, [gridDim.x][4H], one row WRITTEN per workgroup): partial column sums of dgates, whose total is the
// gradient of b_ih (= that of b_hh) — every thread keeps the sums of its four columns of each gate over the rows it
// walks, the workgroup folds its row lanes through LDS and stores 4H floats (a separate reduction pass over dgates would
// read 4H floats per row again; 2048 workgroups adding into the same 4H words with atomics cost more than they saved).
template <int H4>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            const float* __restrict__ dh, const float* dc,
                                                            float* __restrict__ dgates, float* dc_prev,   // (dc_prev may alias dc)
                                                            float* __restrict__ dbias, int R)
{
    constexpr int RL = 256 / H4;                                  // row lanes of a workgroup
    __shared__ f32x4 part[4][256];
    const int k = threadIdx.x % H4, rl = threadIdx.x / H4;
    f32x4 si = { 0.f, 0.f, 0.f, 0.f }, sf = si, sg = si, so = si;
    for (int row = blockIdx.x * RL + rl; row < R; row += gridDim.x * RL) {
        const f32x4* g = reinterpret_cast<const f32x4*>(gates + (size_t)row * 16 * H4);
        const f32x4 gi = g[k], gf = g[H4 + k], gg = g[2 * H4 + k], go = g[3 * H4 + k];
        const size_t rk = (size_t)row * H4 + k;
        const f32x4 c0 = reinterpret_cast<const f32x4*>(c_prev)[rk];
        const f32x4 dhv = reinterpret_cast<const f32x4*>(dh)[rk];
        f32x4 dcv = { 0.f, 0.f, 0.f, 0.f };
        if (dc) dcv = reinterpret_cast<const f32x4*>(dc)[rk];
        f32x4 di, df, dg, dO, dcp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float i = fast_sigmoid(gi[q]), f = fast_sigmoid(gf[q]), gt = fast_tanh(gg[q]), o = fast_sigmoid(go[q]);
            const float c1 = f * c0[q] + i * gt;
            const float tc = fast_tanh(c1);
            const float dct = dcv[q] + dhv[q] * o * (1.0f - tc * tc);
            di[q] = dct * gt * i * (1.0f - i);
            df[q] = dct * c0[q] * f * (1.0f - f);
            dg[q] = dct * i * (1.0f - gt * gt);
            dO[q] = dhv[q] * tc * o * (1.0f - o);
            dcp[q] = dct * f;
        }
        f32x4* dgp = reinterpret_cast<f32x4*>(dgates + (size_t)row * 16 * H4);
        dgp[k] = di;
        dgp[H4 + k] = df;
        dgp[2 * H4 + k] = dg;
        dgp[3 * H4 + k] = dO;
        reinterpret_cast<f32x4*>(dc_prev)[rk] = dcp;
        si += di;
        sf += df;
        sg += dg;
        so += dO;
    }
    if (!dbias) return;
    part[0][threadIdx.x] = si;
    part[1][threadIdx.x] = sf;
    part[2][threadIdx.x] = sg;
    part[3][threadIdx.x] = so;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * H4; idx += 256) {       // (gate, k): fold the row lanes
        const int gate = idx / H4, kk = idx - gate * H4;
        f32x4 acc = part[gate][kk];
        for (int r = 1; r < RL; ++r) acc += part[gate][r * H4 + kk];
        reinterpret_cast<f32x4*>(dbias + (size_t)blockIdx.x * 16 * H4)[gate * H4 + kk] = acc;
    }
}

// Action heads + value head + log_softmax (comm.py:228,239) in one pass over h: out[row][:] =
// [log_softmax(W_0 h + b_0) | log_softmax(W_1 h + b_1) | ... | w_v h + b_v], OT = sum A_k + 1 <= 16 columns.
// 8 lanes per row (each lane a strided set of float4 chunks of the row), W staged in LDS, 3-step shuffle reduce.
#define IC3_MAX_OT 16
__global__ __launch_bounds__(256) void policy_heads_kernel(const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ W, const float* __restrict__ b,
                                                           float* __restrict__ out, int R, int H4, int OT, int nheads,
                                                           int a0, int a1, int a2, int a3)
{
    IC3_DYNAMIC_LDS(float, sW);  // [OT][4*H4]
    for (int i = threadIdx.x; i < OT * H4; i += blockDim.x)
        reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(W)[i];
    __syncthreads();
    const int row = blockIdx.x * (blockDim.x / 8) + (threadIdx.x >> 3);
    const int l8 = threadIdx.x & 7;
    float acc[IC3_MAX_OT];
#pragma unroll
    for (int o = 0; o < IC3_MAX_OT; ++o) acc[o] = 0.0f;
    if (row < R) {
        for (int k = l8; k < H4; k += 8) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(h + (size_t)row * ldh + 4 * k);
#pragma unroll
            for (int o = 0; o < IC3_MAX_OT; ++o) {
                if (o < OT) {
                    const f32x4 w = reinterpret_cast<const f32x4*>(sW)[o * H4 + k];
                    acc[o] += hv.x * w.x + hv.y * w.y + hv.z * w.z + hv.w * w.w;
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < IC3_MAX_OT; ++o) {
        acc[o] += __shfl_xor(acc[o], 1);
        acc[o] += __shfl_xor(acc[o], 2);
        acc[o] += __shfl_xor(acc[o], 4);
    }
    if (row >= R || l8 != 0) return;
    const int sizes[4] = { a0, a1, a2, a3 };
    float* orow = out + (size_t)row * OT;
    int off = 0;
    for (int hd = 0; hd < nheads; ++hd) {
        const int A = sizes[hd];
        float mx = -INFINITY;
#pragma unroll
        for (int o = 0; o < IC3_MAX_OT; ++o)
            if (o >= off && o < off + A) { acc[o] += b[o]; mx = fmaxf(mx, acc[o]); }
        float sum = 0.0f;
#pragma unroll
        for (int o = 0; o < IC3_MAX_OT; ++o)
            if (o >= off && o < off + A) sum += expf(acc[o] - mx);
        const float lse = mx + logf(sum);
#pragma unroll
        for (int o = 0; o < IC3_MAX_OT; ++o)
            if (o >= off && o < off + A) orow[o] = acc[o] - lse;
        off += A;
    }
#pragma unroll
    for (int o = 0; o < IC3_MAX_OT; ++o)
        if (o == off) orow[o] = acc[o] + b[o];  // value head
}

// lstm_cell_kernel + policy_heads_kernel + sample_actions_env_kernel in one pass: the H/4 lanes that produce a row of
// h' also hold it in registers, so the head / value dot products are a lane-local FMA block plus a log2(H/4)-step
// shuffle reduce; lane 0 of the row finishes log_softmax (same arithmetic as policy_heads_kernel) and, when an env
// handle is given, draws the actions of every head (same arithmetic and Philox counters as sample_actions_env_kernel).
// Saves re-reading h (R*H*4 bytes) and three launches per step.  H/4 must be a power of two <= 64.
// (group_sum<G>: env_device.hpp)
template <int H4, int MAXO>
__global__ __launch_bounds__(256) void lstm_cell_heads_kernel(const float* __restrict__ gates, float* __restrict__ c,
                                                              float* __restrict__ h_out, int ldh, int R,
                                                              const float* __restrict__ W, const float* __restrict__ b,
                                                              float* __restrict__ out, int OT, int nheads, int a0, int a1,
                                                              int a2, int a3, const int32_t* __restrict__ episode,
                                                              const int32_t* __restrict__ tstep, uint32_t seed,
                                                              uint32_t gid0, int N, int32_t* __restrict__ action)
{
    IC3_DYNAMIC_LDS(float, sW);  // [OT][4*H4] weights, then [RPB][OT] logits
    constexpr int RPB = 256 / H4;                               // rows per workgroup
    float* sZ = sW + OT * 4 * H4;
    for (int i = threadIdx.x; i < OT * H4; i += blockDim.x)
        reinterpret_cast<f32x4*>(sW)[i] = reinterpret_cast<const f32x4*>(W)[i];
    __syncthreads();
    const int lrow = (int)threadIdx.x / H4, k = (int)threadIdx.x % H4;
    const int row = blockIdx.x * RPB + lrow;
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.0f;
    if (row < R) {
        const f32x4* g = reinterpret_cast<const f32x4*>(gates + (size_t)row * 16 * H4);
        const f32x4 gi = g[k], gf = g[H4 + k], gg = g[2 * H4 + k], go = g[3 * H4 + k];
        f32x4* cp = reinterpret_cast<f32x4*>(c + (size_t)row * 4 * H4) + k;
        const f32x4 c0 = *cp;
        f32x4 c1, h1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c1[q] = fast_sigmoid(gf[q]) * c0[q] + fast_sigmoid(gi[q]) * fast_tanh(gg[q]);
            h1[q] = fast_sigmoid(go[q]) * fast_tanh(c1[q]);
        }
        *cp = c1;
        *reinterpret_cast<f32x4*>(h_out + (size_t)row * ldh + 4 * k) = h1;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            if (o < OT) {
                const f32x4 w = reinterpret_cast<const f32x4*>(sW)[o * H4 + k];
                acc[o] = h1.x * w.x + h1.y * w.y + h1.z * w.z + h1.w * w.w;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        if (o < OT) {
            acc[o] = group_sum<H4>(acc[o]);
            if (k == 0) sZ[lrow * OT + o] = acc[o] + b[o];
        }
    }
    __syncthreads();
    // tail: one task per (row, head) — log_softmax + draw — and one per row for the value head.  RPB * (nheads + 1) can
    // exceed the 256 threads for small H (H = 8 with two heads: 128 rows x 3 tasks), hence the loop.
    const int sizes[4] = { a0, a1, a2, a3 };
    for (int task = threadIdx.x; task < RPB * (nheads + 1); task += blockDim.x) {
        const int tr = task / (nheads + 1), hd = task - tr * (nheads + 1);
        const int grow = blockIdx.x * RPB + tr;
        if (grow >= R) continue;
        const float* z = sZ + tr * OT;
        float* orow = out + (size_t)grow * OT;
        int off = 0;
        for (int i = 0; i < hd && i < nheads; ++i) off += sizes[i];
        if (hd == nheads) {                     // value head (last column)
            orow[off] = z[off];
            continue;
        }
        const int A = sizes[hd];
        float mx = -INFINITY;
        for (int o = 0; o < A; ++o) mx = fmaxf(mx, z[off + o]);
        float sum = 0.0f;                       // hardware exp2 / log2 (~1 ulp): |error| of a log-prob ~1e-7, bar 1e-5
        for (int o = 0; o < A; ++o) sum += __builtin_amdgcn_exp2f(1.4426950408889634f * (z[off + o] - mx));
        const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(sum);
        for (int o = 0; o < A; ++o) orow[off + o] = z[off + o] - lse;
        if (action) {
            const int e = grow / N, n = grow - e * N;
            const uint32_t x = philox_x24(seed, gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)episode[e], (uint32_t)tstep[e],
                                          (uint32_t)(hd * N + n));
            const float u = (float)x * (1.0f / 16777216.0f);
            float cdf = 0.0f;
            int a = A - 1;
            for (int o = 0; o < A - 1; ++o) {
                cdf += expf(z[off + o] - lse);
                if (u < cdf) {
                    a = o;
                    break;
                }
            }
            action[(size_t)hd * R + grow] = a;
        }
    }
}

// action_utils.py:32-36: torch.multinomial(exp(logp), 1) per row.  Inverse-CDF on the injected uniform:
// first a with u < sum_{b<=a} exp(logp_b), last action as fallback (fp32, left-to-right).
__global__ __launch_bounds__(256) void sample_actions_kernel(const float* __restrict__ logp, int ld, int A, int head,
                                                             uint32_t seed, uint32_t gid0, uint32_t episode, uint32_t t,
                                                             int32_t* __restrict__ action,
                                                             float* __restrict__ chosen_logp, int E, int N)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= E * N) return;
    const int e = row / N, n = row - e * N;
    const uint32_t x = philox_x24(seed, gid0 + (uint32_t)e, DOMAIN_SAMPLE, episode, t, (uint32_t)(head * N + n));
    const float u = (float)x * (1.0f / 16777216.0f);
    const float* lp = logp + (size_t)row * ld;
    float cdf = 0.0f;
    int a = A - 1;
    for (int b = 0; b < A - 1; ++b) {
        cdf += expf(lp[b]);
        if (u < cdf) {
            a = b;
            break;
        }
    }
    action[row] = a;
    if (chosen_logp) chosen_logp[row] = lp[a];
}

// Same draw, but (episode, t) come from the env handle's device-side counters (episode[e], t[e]) instead of
// by-value arguments, so a captured step graph stays valid across steps and episodes.
__global__ __launch_bounds__(256) void sample_actions_env_kernel(const float* __restrict__ logp, int ld, int A, int head,
                                                                 uint32_t seed, uint32_t gid0,
                                                                 const int32_t* __restrict__ episode,
                                                                 const int32_t* __restrict__ tstep,
                                                                 int32_t* __restrict__ action,
                                                                 float* __restrict__ chosen_logp, int E, int N)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= E * N) return;
    const int e = row / N, n = row - e * N;
    const uint32_t x = philox_x24(seed, gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)episode[e], (uint32_t)tstep[e],
                                  (uint32_t)(head * N + n));
    const float u = (float)x * (1.0f / 16777216.0f);
    const float* lp = logp + (size_t)row * ld;
    float cdf = 0.0f;
    int a = A - 1;
    for (int b = 0; b < A - 1; ++b) {
        cdf += expf(lp[b]);
        if (u < cdf) {
            a = b;
            break;
        }
    }
    action[row] = a;
    if (chosen_logp) chosen_logp[row] = lp[a];
}

int sample_actions_env(const ic3_env* env, const float* logp, int ld, int A, int head, int32_t* action, float* chosen_logp,
                       hipStream_t s)
{
    const int E = env->dims.E, N = env->dims.N, rows = E * N;
    if (ld <= 0) ld = A;
    const uint32_t seed = env->kind == IC3_ENV_PP ? env->pp.seed : env->tj.seed;
    const uint32_t gid0 = env->kind == IC3_ENV_PP ? env->pp.env_id_offset : env->tj.env_id_offset;
    hipLaunchKernelGGL(sample_actions_env_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, logp, ld, A, head, seed, gid0,
                       env->f("episode"), env->f("t"), action, chosen_logp, E, N);
    IC3_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void random_actions_kernel(int32_t* __restrict__ action, int naction, uint32_t seed,
                                                             uint32_t gid0, uint32_t episode, uint32_t t, int E, int N)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= E * N) return;
    const int e = row / N, n = row - e * N;
    action[row] = (int32_t)scale24(philox_x24(seed, gid0 + (uint32_t)e, DOMAIN_BENCH, episode, t, (uint32_t)n),
                                   (uint32_t)naction);
}

}  // namespace ic3

static int comm_masked_mean_launch(const float* h, int ldh, const int32_t* alive, const int32_t* comm_action, const float* addend,
                                   int lda, float* out, int E, int N, int H, int mode_avg, int mask_self, ic3_stream stream,
                                   const char* who, const float* out_scale = nullptr)
{
    if (!h || !out || E <= 0 || N <= 0 || H <= 0) return ic3::fail(-22, std::string(who) + ": bad arguments");
    if (ldh <= 0) ldh = H;
    if (lda <= 0) lda = H;
    if ((H & 3) == 0 && (ldh & 3) == 0 && (lda & 3) == 0 && H / 4 <= 256) {
        const int H4 = H / 4, per_block = 256 / H4;
        const dim3 grid((E + per_block - 1) / per_block);
        hipStream_t s = (hipStream_t)stream;
        if (N <= 16)
            hipLaunchKernelGGL(ic3::comm_masked_mean_kernel<16>, grid, dim3(256), 0, s, h, ldh, alive, comm_action, out, E, N,
                               H4, mode_avg, mask_self, addend, lda, out_scale);
        else if (N <= 32)
            hipLaunchKernelGGL(ic3::comm_masked_mean_kernel<32>, grid, dim3(256), 0, s, h, ldh, alive, comm_action, out, E, N,
                               H4, mode_avg, mask_self, addend, lda, out_scale);
        else
            hipLaunchKernelGGL(ic3::comm_masked_mean_kernel<0>, grid, dim3(256), 0, s, h, ldh, alive, comm_action, out, E, N,
                               H4, mode_avg, mask_self, addend, lda, out_scale);
    } else {
        if (addend || out_scale) return ic3::fail(-38, std::string(who) + ": H, ldh and lda must be multiples of 4");
        const int threads = H >= 256 ? 256 : ((H + 63) / 64) * 64;
        hipLaunchKernelGGL(ic3::comm_masked_mean_scalar_kernel, dim3(E), dim3(threads), 0, (hipStream_t)stream, h, ldh,
                           alive, comm_action, out, N, H, mode_avg, mask_self);
    }
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_comm_masked_mean(const float* h, int ldh, const int32_t* alive, const int32_t* comm_action, float* out,
                                    int E, int N, int H, int mode_avg, int mask_self, ic3_stream stream)
{
    return comm_masked_mean_launch(h, ldh, alive, comm_action, nullptr, 0, out, E, N, H, mode_avg, mask_self, stream,
                                   "ic3_comm_masked_mean");
}

extern "C" int ic3_comm_masked_mean_add(const float* h, int ldh, const int32_t* alive, const int32_t* comm_action,
                                        const float* addend, int lda, const float* out_row_scale, float* out, int E, int N, int H,
                                        int mode_avg, int mask_self, ic3_stream stream)
{
    if (!addend) return ic3::fail(-22, "ic3_comm_masked_mean_add: null addend");
    return comm_masked_mean_launch(h, ldh, alive, comm_action, addend, lda, out, E, N, H, mode_avg, mask_self, stream,
                                   "ic3_comm_masked_mean_add", out_row_scale);
}

extern "C" int ic3_lstm_cell(const float* gates, float* c, float* h_out, int ldh, int R, int H, ic3_stream stream)
{
    if (!gates || !c || !h_out || R <= 0 || H <= 0 || (H & 3) || (ldh & 3) || ldh < H)
        return ic3::fail(-22, "ic3_lstm_cell: bad arguments (H and ldh must be multiples of 4)");
    const long long n = (long long)R * (H / 4);
    hipLaunchKernelGGL(ic3::lstm_cell_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates,
                       c, h_out, ldh, R, H / 4);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_lstm_cell_backward(const float* gates, const float* c_prev, const float* dh, const float* dc, float* dgates,
                                      float* dc_prev, float* dbias, int R, int H, ic3_stream stream)
{
    if (!gates || !c_prev || !dh || !dgates || !dc_prev || R <= 0 || H <= 0 || (H & 3))
        return ic3::fail(-22, "ic3_lstm_cell_backward: bad arguments (H must be a multiple of 4)");
    const int H4 = H / 4;
    if (H4 > 64 || (H4 & (H4 - 1))) return ic3::fail(-38, "ic3_lstm_cell_backward: H/4 must be a power of two <= 64");
    const int RL = 256 / H4;
    int blocks = (R + RL - 1) / RL;
    if (blocks > IC3_LSTM_BWD_MAX_PARTIALS) blocks = IC3_LSTM_BWD_MAX_PARTIALS;   // grid-stride: one partial row per workgroup
    hipStream_t s = (hipStream_t)stream;
#define IC3_LCB(N) hipLaunchKernelGGL(ic3::lstm_cell_bwd_kernel<N>, dim3(blocks), dim3(256), 0, s, gates, c_prev, dh, dc, dgates, dc_prev, dbias, R)
    switch (H4) {
    case 1: IC3_LCB(1); break;
    case 2: IC3_LCB(2); break;
    case 4: IC3_LCB(4); break;
    case 8: IC3_LCB(8); break;
    case 16: IC3_LCB(16); break;
    case 32: IC3_LCB(32); break;
    default: IC3_LCB(64); break;
    }
#undef IC3_LCB
    IC3_HIP(hipGetLastError());
    return blocks;   // rows of `dbias` written
}

extern "C" int ic3_lstm_cell_heads(const float* gates, float* c, float* h_out, int ldh, int R, int H, const float* W,
                                   const float* b, const int32_t* head_sizes, int nheads, float* out, const ic3_env* env,
                                   int32_t* action, ic3_stream stream)
{
    if (!gates || !c || !h_out || !W || !b || !head_sizes || !out || R <= 0 || H <= 0 || (H & 3) || (ldh & 3) || ldh < H ||
        nheads < 1 || nheads > 4)
        return ic3::fail(-22, "ic3_lstm_cell_heads: bad arguments (1..4 heads, H and ldh multiples of 4)");
    const int H4 = H / 4;
    if (H4 > 64 || (H4 & (H4 - 1))) return ic3::fail(-38, "ic3_lstm_cell_heads: H/4 must be a power of two <= 64");
    int sz[4] = { 0, 0, 0, 0 }, OT = 1;
    for (int i = 0; i < nheads; ++i) { sz[i] = head_sizes[i]; OT += head_sizes[i]; }
    if (OT > IC3_MAX_OT) return ic3::fail(-22, "ic3_lstm_cell_heads: more than 15 actions in total");
    const int32_t *ep = nullptr, *ts = nullptr;
    uint32_t seed = 0, gid0 = 0;
    int N = 1;
    if (action) {
        if (!env || env->dims.E * env->dims.N != R)
            return ic3::fail(-22, "ic3_lstm_cell_heads: sampling needs the env handle whose E*N equals R");
        ep = env->f("episode");
        ts = env->f("t");
        seed = env->kind == IC3_ENV_PP ? env->pp.seed : env->tj.seed;
        gid0 = env->kind == IC3_ENV_PP ? env->pp.env_id_offset : env->tj.env_id_offset;
        N = env->dims.N;
    }
    const int rpb = 256 / H4;
    const size_t lds = ((size_t)OT * H + (size_t)rpb * OT) * sizeof(float);
    const dim3 grid((R + rpb - 1) / rpb), block(256);
    hipStream_t s = (hipStream_t)stream;
#define IC3_LCH(h4)                                                                                                      \
    case h4:                                                                                                             \
        if (OT <= 8)                                                                                                     \
            hipLaunchKernelGGL((ic3::lstm_cell_heads_kernel<h4, 8>), grid, block, lds, s, gates, c, h_out, ldh, R, W, b,  \
                               out, OT, nheads, sz[0], sz[1], sz[2], sz[3], ep, ts, seed, gid0, N, action);               \
        else                                                                                                             \
            hipLaunchKernelGGL((ic3::lstm_cell_heads_kernel<h4, IC3_MAX_OT>), grid, block, lds, s, gates, c, h_out, ldh,  \
                               R, W, b, out, OT, nheads, sz[0], sz[1], sz[2], sz[3], ep, ts, seed, gid0, N, action);      \
        break;
    switch (H4) {
        IC3_LCH(1) IC3_LCH(2) IC3_LCH(4) IC3_LCH(8) IC3_LCH(16) IC3_LCH(32) IC3_LCH(64)
    }
#undef IC3_LCH
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_policy_heads(const float* h, int ldh, const float* W, const float* b, const int32_t* head_sizes,
                                int nheads, float* out, int R, int H, ic3_stream stream)
{
    if (!h || !W || !b || !head_sizes || !out || R <= 0 || H <= 0 || (H & 3) || (ldh & 3) || nheads < 1 || nheads > 4)
        return ic3::fail(-22, "ic3_policy_heads: bad arguments (1..4 heads, H % 4 == 0)");
    int sz[4] = { 0, 0, 0, 0 }, OT = 1;
    for (int i = 0; i < nheads; ++i) { sz[i] = head_sizes[i]; OT += head_sizes[i]; }
    if (OT > IC3_MAX_OT) return ic3::fail(-22, "ic3_policy_heads: more than 15 actions in total");
    const size_t lds = (size_t)OT * H * sizeof(float);
    hipLaunchKernelGGL(ic3::policy_heads_kernel, dim3((R + 31) / 32), dim3(256), lds, (hipStream_t)stream, h, ldh, W, b,
                       out, R, H / 4, OT, nheads, sz[0], sz[1], sz[2], sz[3]);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_sample_actions(const float* logp, int ld, int A, int head, uint32_t seed, uint32_t env_id_offset,
                                  uint32_t episode, uint32_t t, int32_t* action, float* chosen_logp, int E, int N,
                                  ic3_stream stream)
{
    if (!logp || !action || A <= 0 || E <= 0 || N <= 0) return ic3::fail(-22, "ic3_sample_actions: bad arguments");
    if (ld <= 0) ld = A;
    const int rows = E * N;
    hipLaunchKernelGGL(ic3::sample_actions_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, logp, ld,
                       A, head, seed, env_id_offset, episode, t, action, chosen_logp, E, N);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_random_actions(int32_t* action, int naction, uint32_t seed, uint32_t env_id_offset, uint32_t episode,
                                  uint32_t t, int E, int N, ic3_stream stream)
{
    if (!action || naction <= 0 || E <= 0 || N <= 0) return ic3::fail(-22, "ic3_random_actions: bad arguments");
    const int rows = E * N;
    hipLaunchKernelGGL(ic3::random_actions_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, action,
                       naction, seed, env_id_offset, episode, t, E, N);
    IC3_HIP(hipGetLastError());
    return 0;
}


namespace ic3 {

// ---- ic3_heads_grad: the weight / bias gradient of the heads + value head over a WHOLE episode in one pass ------------------
// (trainer.py:128-225 through comm.py:228,239): dW [OT][H] += sum_m d[m][o] h[m][c], db [OT] += sum_m d[m][o], m over all
// M = T * R (step, row) pairs.  Per step this was a K = R library GEMM with an 8 x 128 result (+ two reduction launches);
// as ONE product its K = 6.5 M (PP-hard) is a shape the library tunes for minutes.  HBM-bound: every h row is read once
// (M * H * 4 bytes: 3.4 GB per PP-hard update = 0.6 ms).  Round 6: on the fp32 matrix instruction — v_mfma_f32_32x32x2_f32 with
// A = d^T (output row = o, the 32 - OT rows beyond the heads are zeros), B = h, K = two rows per instruction: a lane loads ONE
// float of h (its column of the wave's 32-column block, row by lane half) and one of d per instruction, nothing goes through
// LDS (rounds 4-5 broadcast the d rows from LDS to one thread per column: 1.78 ms, bound by the LDS instruction issue).  A wave
// owns a 32-column block; with fewer than four blocks (hid 64) the waves split the rows too.  Partials [parts][17][H] (row 16:
// the bias sums) are reduced by a second small launch in a fixed order (reproducible).
typedef float hg_f32x16 __attribute__((ext_vector_type(16)));
template <int H>
__global__ __launch_bounds__(256) void heads_grad_kernel(const float* __restrict__ d, const float* __restrict__ h, long long M, int OT,
                                                         float* __restrict__ partial /* [grid * NRG][17][H] */)
{
    constexpr int CB = H / 32, NRG = CB >= 4 ? 1 : 4 / CB, BPW = CB > 4 ? CB / 4 : 1, U = 8;   // column blocks, row groups, blocks per wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int cb0 = (w % (CB < 4 ? CB : 4)) * BPW, rg = CB >= 4 ? 0 : w / CB;
    const long long per = (M + gridDim.x - 1) / gridDim.x;
    const long long m0 = (long long)blockIdx.x * per;
    long long nrows = M - m0 < per ? M - m0 : per;
    if (nrows < 0) nrows = 0;
    const uint32_t hbytes = (uint32_t)(nrows * H * 4), dbytes = (uint32_t)(nrows * OT * 4);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h + m0 * H), 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d + m0 * OT), 0, dbytes, 0x00020000);
    hg_f32x16 acc[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.0f;
    float bsum = 0.0f;
    const bool has_o = li < OT;
    const int hv0 = ((2 * rg + lh) * H + 32 * cb0 + li) * 4, dv0 = ((2 * rg + lh) * OT + (has_o ? li : 0)) * 4;
    for (long long base = 0; base < nrows; base += 2 * NRG * U) {
        float hv[U][BPW], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = (int)base + 2 * NRG * u;                // rows r + 2 rg + lh of the workgroup's range (past it: 0)
#pragma unroll
            for (int b = 0; b < BPW; ++b)
                hv[u][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, hv0 + 32 * b * 4, r * H * 4, 0));
            const float dl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, dv0, r * OT * 4, 0));
            dv[u] = has_o ? dl : 0.0f;                           // (output rows beyond the heads: zeros)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int b = 0; b < BPW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[u], hv[u][b], acc[b], 0, 0, 0);
            bsum += dv[u];
        }
    }
    // accumulator register reg, lane (li, lh) <-> o = (reg & 3) + 8 (reg >> 2) + 4 lh, column 32 block + li: o < 16 <-> reg < 8
    float* out = partial + ((size_t)blockIdx.x * NRG + rg) * 17 * H;
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int reg = 0; reg < 8; ++reg) out[((reg & 3) + 8 * (reg >> 2) + 4 * lh) * H + 32 * (cb0 + b) + li] = acc[b][reg];
    bsum += __shfl_xor(bsum, 32);
    if (cb0 == 0 && lh == 0 && li < 16) out[16 * H + li] = bsum;      // (li >= OT: 0)
}

// dW / db += the workgroups' partials.  16 outputs x 16 shares of the partials per workgroup (a thread walking all <= 1024
// partials 8.7 KB apart is a latency chain: 417 us at hid 128 — round 6), the shares folded in share order: reproducible.
__global__ __launch_bounds__(256) void heads_grad_reduce_kernel(const float* __restrict__ partial, int nparts, int H, int OT,
                                                                float* __restrict__ dW, float* __restrict__ db)
{
    __shared__ float sh[16][17];
    const int ol = threadIdx.x & 15, share = threadIdx.x >> 4, i = blockIdx.x * 16 + ol, n = OT * H + OT;
    const int per = (nparts + 15) / 16, p0 = share * per, p1 = min(nparts, p0 + per);
    const int off = i < OT * H ? i : 16 * H + (i - OT * H);
    float v = 0.0f;
    if (i < n)
        for (int p = p0; p < p1; ++p) v += partial[(size_t)p * 17 * H + off];
    sh[share][ol] = v;
    __syncthreads();
    if (share == 0 && i < n) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][ol];
        if (i < OT * H) dW[i] += t;
        else db[i - OT * H] += t;
    }
}

}  // namespace ic3

extern "C" size_t ic3_heads_grad_scratch_floats(int H) { return (size_t)1024 * 17 * (size_t)H; }

extern "C" int ic3_heads_grad(const float* d, const float* h, long long M, int H, int OT, float* dW, float* db, float* scratch,
                              ic3_stream stream)
{
    using namespace ic3;
    if (!d || !h || !dW || !db || !scratch || M <= 0 || OT < 1 || OT > 16) return fail(-22, "ic3_heads_grad: bad arguments");
    if (H != 64 && H != 128 && H != 256) return fail(-38, "ic3_heads_grad: hid_size 64 / 128 / 256");
    hipStream_t s = (hipStream_t)stream;
    const int nrg = H == 64 ? 2 : 1;                              // (hid 64: two column blocks, the waves split the rows as well)
    int grid = (int)std::min<long long>(1024 / nrg, (M + 63) / 64);
    if ((M + grid - 1) / grid * (long long)H * 4 >= (1ll << 31)) return fail(-22, "ic3_heads_grad: too many rows per workgroup");
    if (H == 64) hipLaunchKernelGGL((heads_grad_kernel<64>), dim3(grid), dim3(256), 0, s, d, h, M, OT, scratch);
    else if (H == 128) hipLaunchKernelGGL((heads_grad_kernel<128>), dim3(grid), dim3(256), 0, s, d, h, M, OT, scratch);
    else hipLaunchKernelGGL((heads_grad_kernel<256>), dim3(grid), dim3(256), 0, s, d, h, M, OT, scratch);
    IC3_HIP(hipGetLastError());
    const int n = OT * H + OT;
    hipLaunchKernelGGL(heads_grad_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, s, scratch, grid * nrg, H, OT, dW, db);
    IC3_HIP(hipGetLastError());
    return 0;
}
