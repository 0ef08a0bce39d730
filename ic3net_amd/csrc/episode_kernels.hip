// ic3_episode_finalize — what get_episode derives per step next to the policy / env calls
// (/root/reference/trainer.py:70-105,109-110): the live / alive / episode masks of every transition and the
// per-agent reward and comm-action sums, for n lock-step slots of E envs, as ONE launch over the episode buffers the
// step launches wrote ([T][E] / [T][E][N], step-major).  One thread per (env, agent) walks the n slots (its env's
// `done` history is a running product), so every access is coalesced across the (env, agent) index; the sums are
// reduced in a fixed order (block partials, then a one-block launch over them), so the statistics are reproducible.
#include <hip/hip_runtime.h>

#include "ic3_common.hpp"

namespace ic3 {

constexpr int EF_THREADS = 256;

// stats layout: [0] num_steps (sum of live), [1] envs whose last held slot has done set, [2, 2+N) reward sums,
// [2+N, 2+2N) gate (comm_action) sums
// INFO: alive / is_completed present (TJ); GATE: talk-head actions present — compile-time so that every load of a batch
// is unconditional (a load under a branch is followed by its own s_waitcnt: one round trip per load)
template <bool INFO, bool GATE>
__global__ __launch_bounds__(EF_THREADS) void episode_finalize_kernel(ic3_episode a, int epb /* envs per block */)
{
    __shared__ double sh[EF_THREADS][2];
    __shared__ double shn[EF_THREADS][2];
    const int tid = threadIdx.x, N = a.N, E = a.E, n = a.n;
    const int el = tid / N, ag = tid - el * N;
    const int e = blockIdx.x * epb + el;
    const bool valid = el < epb && e < E;
    double rsum = 0.0, gsum = 0.0, steps = 0.0, zero_len = 0.0;
    if (valid) {
        const size_t EN = (size_t)E * N;
        const size_t en = (size_t)e * N + ag;
        float live = 1.0f;
        // slots in batches of TB: all loads of a batch are issued before its first store (the compiler cannot move a
        // load above a store through these pointers, and one slot per round trip made the launch latency-bound: 218 us
        // for 20 x 8192 x 10 transitions)
        constexpr int TB = 5;
        for (int t0 = 0; t0 < n; t0 += TB) {
            int dn[TB], alv[TB], cmp[TB], gt[TB];
            float rw[TB];
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int tc = min(t0 + k, n - 1);                 // slots past the end re-read the last one (ignored)
                const size_t i = (size_t)tc * EN + en;
                dn[k] = a.done[(size_t)tc * E + e];
                alv[k] = 1;
                cmp[k] = 0;
                gt[k] = 0;
                if constexpr (INFO) {
                    alv[k] = a.alive[i];
                    cmp[k] = a.is_completed[i];
                }
                rw[k] = a.reward[i];
                if constexpr (GATE) gt[k] = a.gate[(size_t)tc * a.gate_stride + en];
            }
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int t = t0 + k;
                if (t >= n) break;
                const bool d = dn[k] != 0;
                const bool done_t = d || (a.forced_last && t == n - 1);            // trainer.py:90
                const size_t i = (size_t)t * EN + en;
                a.alive_mask[i] = (float)alv[k] * live;                             // trainer.py:78-81
                float mini = 1.0f;                                                  // trainer.py:98-99 (only when not done, Q26)
                if (INFO && !done_t) mini = 1.0f - (float)cmp[k];
                a.episode_mini_mask[i] = mini;
                rsum += (double)rw[k];                                              // trainer.py:86
                if (a.gate_ones) gsum += (double)live;                              // trainer.py:73-75
                else if (GATE) gsum += (double)((float)gt[k] * live);
                if (ag == 0) {
                    a.live[(size_t)t * E + e] = live;
                    a.episode_mask[(size_t)t * E + e] = done_t ? 0.0f : 1.0f;        // trainer.py:92-96
                    steps += (double)live;                                          // trainer.py:109-110
                }
                if (!a.auto_reset && d) live = 0.0f;     // lock-step: a finished env idles (auto-reset: every slot is real)
            }
        }
        if (ag == 0) {
            const bool dl = n > 0 && a.done[(size_t)(n - 1) * E + e] != 0;
            // live[n-1] * (1 - done[n-1]); lock-step: the running product already holds it
            a.live_after[e] = a.auto_reset ? (dl ? 0.0f : 1.0f) : live;
            zero_len = dl ? 1.0 : 0.0;
        }
    }
    sh[tid][0] = rsum;
    sh[tid][1] = gsum;
    shn[tid][0] = 0.0;
    shn[tid][1] = 0.0;
    __syncthreads();
    if (valid && ag == 0) {
        shn[el][0] = steps;
        shn[el][1] = zero_len;
    }
    __syncthreads();
    double* part = a.scratch + (size_t)blockIdx.x * (2 + 2 * N);
    if (tid < N) {                                   // fixed order over the block's envs
        double r = 0.0, g = 0.0;
        for (int k = 0; k < epb; ++k) {
            r += sh[k * N + tid][0];
            g += sh[k * N + tid][1];
        }
        part[2 + tid] = r;
        part[2 + N + tid] = g;
    }
    if (tid == 0) {
        double s = 0.0, z = 0.0;
        for (int k = 0; k < epb; ++k) {
            s += shn[k][0];
            z += shn[k][1];
        }
        part[0] = s;
        part[1] = z;
    }
}

// Second launch (one block): a kernel boundary publishes the block partials; a device-scope fence per block inside the
// first launch wrote back the whole L2 each time (the masks it had just stored) and made it 5x slower.
__global__ __launch_bounds__(EF_THREADS) void episode_reduce_kernel(ic3_episode a, int nblocks)
{
    __shared__ double sh[EF_THREADS][2];
    const int tid = threadIdx.x, N = a.N;
    // fixed-order sum over the blocks: the block range is cut into `nseg` segments summed by different threads with
    // several loads in flight (one thread per value walking all blocks took ~200 us of serialised load latency),
    // then the segment sums are added in order
    const int KK = 2 + 2 * N;
    const int nseg = KK <= EF_THREADS ? EF_THREADS / KK : 1;
    const int per = (nblocks + nseg - 1) / nseg;
    double* segsum = &sh[0][0];                              // [nseg][KK] when KK <= EF_THREADS (2 * EF_THREADS doubles)
    for (int k0 = 0; k0 < KK; k0 += EF_THREADS) {
        const int k = k0 + (KK <= EF_THREADS ? tid % KK : tid), seg = KK <= EF_THREADS ? tid / KK : 0;
        double v = 0.0;
        if (k < KK && seg < nseg) {
            const int b0 = seg * per, b1 = min(nblocks, b0 + per);
            constexpr int U = 8;
            for (int b = b0; b < b1; b += U) {
                double x[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    x[u] = (b + u < b1) ? __builtin_nontemporal_load(a.scratch + (size_t)(b + u) * KK + k) : 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) v += x[u];
            }
        }
        if (KK <= EF_THREADS) {
            __syncthreads();
            if (seg < nseg) segsum[seg * KK + k] = v;
            __syncthreads();
            if (tid < KK) {
                double t = 0.0;
                for (int sg = 0; sg < nseg; ++sg) t += segsum[sg * KK + tid];
                a.stats[tid] = t;
            }
        } else if (k < KK) {
            a.stats[k] = v;
        }
    }
}

// ic3_returns_scan — the reversed return scan of compute_grad (/root/reference/trainer.py:162-171) over T slots of E envs x N
// agents in ONE launch (as a Python loop over the slots it is ~9 tensor ops per slot: 720 launches per 80-step update):
//   coop[t]  = reward[t] + gamma * coop[t + 1]  * episode_mask[t]
//   ncoop[t] = reward[t] + gamma * ncoop[t + 1] * episode_mask[t] * episode_mini_mask[t]
//   returns[t][e][n] = mean_ratio * mean_n coop[t][e][:] + (1 - mean_ratio) * ncoop[t][e][n]
// One thread per (env, agent) walks the slots backwards (coalesced across the (env, agent) index); the mean over an env's
// agents goes through LDS (the block holds whole envs).  Slots are read in batches so that several loads are in flight.
__global__ __launch_bounds__(EF_THREADS) void returns_scan_kernel(const float* __restrict__ reward, const float* __restrict__ emask,
                                                                  const float* __restrict__ mini, float gamma, float mean_ratio,
                                                                  float* __restrict__ returns, int T, int E, int N, int epb)
{
    __shared__ float sh[EF_THREADS];
    const int tid = threadIdx.x;
    const int el = tid / N, ag = tid - el * N;
    const int e = blockIdx.x * epb + el;
    const bool valid = el < epb && e < E;
    const size_t EN = (size_t)E * N, en = valid ? (size_t)e * N + ag : 0;
    const int ec = valid ? e : 0;
    const float invN = 1.0f / (float)N;
    float coop = 0.0f, ncoop = 0.0f;
    constexpr int TB = 4;
    for (int t1 = T; t1 > 0; t1 -= TB) {
        float rw[TB], em[TB], mm[TB];
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int t = max(t1 - 1 - k, 0);
            rw[k] = reward[(size_t)t * EN + en];
            em[k] = emask[(size_t)t * E + ec];
            mm[k] = mini[(size_t)t * EN + en];
        }
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int t = t1 - 1 - k;
            if (t < 0) break;                                       // (uniform)
            coop = rw[k] + gamma * coop * em[k];
            ncoop = rw[k] + gamma * ncoop * em[k] * mm[k];
            float mean = 0.0f;
            if (mean_ratio != 0.0f) {                               // (uniform)
                __syncthreads();
                sh[tid] = valid ? coop : 0.0f;
                __syncthreads();
                const int base = valid ? el * N : 0;                  // (the block's spare threads: el * N + j would leave the array)
                for (int j = 0; j < N; ++j) mean += sh[base + j];     // fixed order
                mean *= invN;
            }
            if (valid) returns[(size_t)t * EN + en] = mean_ratio * mean + (1.0f - mean_ratio) * ncoop;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ic3_loss_gradients — compute_grad's losses and what they hand back to the policy's outputs (/root/reference/trainer.py:173-218),
// one thread per transition row (slot t, agent row r), in ONE launch instead of ~40 elementwise / gather / scatter / reduce
// launches:  adv = (returns - value - shift) * scale  (normalize_rewards: shift / scale from the caller, else 0 / 1);
//   action_loss = sum -adv * (sum_k logp_k[a_k]) * alive        value_loss = sum (value - returns)^2 * alive
//   entropy     = sum_k sum_o -logp_k[o] exp(logp_k[o]) * live  (no alive mask there, trainer.py:211-218)
//   d_out[k][o] = dlp_o - exp(logp_o) * sum_o' dlp_o',  dlp_o = [o == a_k] (-adv * alive) + entr * live * exp(logp_o) (logp_o + 1)
//   d_out[value] = 2 * value_coeff * (value - returns) * alive          (the log-softmax folded in: gradients w.r.t. its INPUT)
// The three sums: per-workgroup partials in double (fixed order inside a workgroup), summed by the caller.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float lg_f32x4 __attribute__((ext_vector_type(4)));
struct LossGradArgs {
    const float* out;        // [T][R][OT] = [log-probs of every head | value]
    const int32_t* action;   // [T][nheads][R]
    const float* returns;    // [T][R]
    const float* alive;      // [T][R]  (already times live)
    const float* live;       // [T][E]
    float* d_out;            // [T][R][OT]
    double* sums;            // [gridDim.x][3]
    long long M;             // T * R
    int R, N, OT, nheads, a0, a1, a2, a3;
    float shift, scale, entr, value_coeff;
};
__global__ __launch_bounds__(256) void loss_gradients_kernel(const LossGradArgs a)
{
    __shared__ double red[3][4];
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    double s_act = 0.0, s_val = 0.0, s_ent = 0.0;
    if (m < a.M) {
        const long long t = m / a.R;
        const int r = (int)(m - t * a.R), e = r / a.N;
        // the row in registers (OT <= 16): 16-byte loads / stores when the row is a whole number of them
        float o[16], d[16];
        const float* og = a.out + m * a.OT;
        float* dg = a.d_out + m * a.OT;
        const bool vec = (a.OT & 3) == 0;
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * q < a.OT) {
                    const lg_f32x4 v = reinterpret_cast<const lg_f32x4*>(og)[q];
                    o[4 * q] = v[0], o[4 * q + 1] = v[1], o[4 * q + 2] = v[2], o[4 * q + 3] = v[3];
                }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < a.OT) o[q] = og[q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.f;
        const float value = o[a.OT - 1], ret = a.returns[m], al = a.alive[m], lv = a.live[t * (a.R / a.N) + e];
        const float adv = (ret - value - a.shift) * a.scale;
        const float w = -adv * al;
        const int sizes[4] = { a.a0, a.a1, a.a2, a.a3 };
        int off = 0;
        float lp_sum = 0.f;
        for (int k = 0; k < a.nheads; ++k) {
            const int A = sizes[k], act = a.action[(t * a.nheads + k) * a.R + r];
            float dsum = 0.f, ent = 0.f;
            for (int j = 0; j < A; ++j) {                         // (A <= 15: the row's logits are read twice, from L1)
                const float lp = o[off + j], p = expf(lp);
                float dl = (j == act) ? w : 0.f;
                if (a.entr > 0.f) dl += a.entr * lv * p * (lp + 1.0f);
                dsum += dl;
                ent -= lp * p;
            }
            for (int j = 0; j < A; ++j) {
                const float lp = o[off + j], p = expf(lp);
                float dl = (j == act) ? w : 0.f;
                if (a.entr > 0.f) dl += a.entr * lv * p * (lp + 1.0f);
                d[off + j] = dl - p * dsum;
            }
            lp_sum += o[off + act];
            s_ent += (double)(ent * lv);
            off += A;
        }
        d[a.OT - 1] = 2.0f * a.value_coeff * (value - ret) * al;
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * q < a.OT) reinterpret_cast<lg_f32x4*>(dg)[q] = lg_f32x4{ d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3] };
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < a.OT) dg[q] = d[q];
        }
        s_act = (double)(-adv * lp_sum * al);
        s_val = (double)((value - ret) * (value - ret) * al);
    }
    // workgroup sums: lanes by shuffles, the four waves through LDS, in a fixed order
    for (int sh = 32; sh > 0; sh >>= 1) {
        s_act += __shfl_xor(s_act, sh);
        s_val += __shfl_xor(s_val, sh);
        s_ent += __shfl_xor(s_ent, sh);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = s_act;
        red[1][wv] = s_val;
        red[2][wv] = s_ent;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        a.sums[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

}  // namespace ic3

using namespace ic3;

extern "C" {

int ic3_loss_gradients_partials(long long T, long long R) { return (T <= 0 || R <= 0) ? 0 : (int)((T * R + 255) / 256); }

int ic3_loss_gradients(const float* out, const int32_t* action, const float* returns, const float* alive_mask, const float* live,
                       const int32_t* head_sizes, int nheads, float adv_shift, float adv_scale, float entr, float value_coeff,
                       float* d_out, double* sums, int T, int E, int N, ic3_stream stream)
{
    Range range("ic3_loss_gradients");
    if (!out || !action || !returns || !alive_mask || !live || !head_sizes || !d_out || !sums) return fail(-22, "ic3_loss_gradients: null argument");
    if (T <= 0 || E <= 0 || N <= 0 || nheads < 1 || nheads > 4) return fail(-22, "ic3_loss_gradients: T, E, N > 0, 1..4 heads");
    LossGradArgs a{};
    int sz[4] = { 0, 0, 0, 0 };
    a.OT = 1;
    for (int k = 0; k < nheads; ++k) {
        sz[k] = head_sizes[k];
        if (sz[k] < 1) return fail(-22, "ic3_loss_gradients: empty action head");
        a.OT += sz[k];
    }
    a.out = out;
    a.action = action;
    a.returns = returns;
    a.alive = alive_mask;
    a.live = live;
    a.d_out = d_out;
    a.sums = sums;
    a.R = E * N;
    a.N = N;
    a.M = (long long)T * a.R;
    a.nheads = nheads;
    a.a0 = sz[0];
    a.a1 = sz[1];
    a.a2 = sz[2];
    a.a3 = sz[3];
    a.shift = adv_shift;
    a.scale = adv_scale;
    a.entr = entr;
    a.value_coeff = value_coeff;
    hipLaunchKernelGGL(loss_gradients_kernel, dim3(ic3_loss_gradients_partials(T, a.R)), dim3(256), 0, (hipStream_t)stream, a);
    IC3_HIP(hipGetLastError());
    return 0;
}

int ic3_returns_scan(const float* reward, const float* episode_mask, const float* episode_mini_mask, float gamma, float mean_ratio,
                     float* returns, int T, int E, int N, ic3_stream stream)
{
    Range range("ic3_returns_scan");
    if (!reward || !episode_mask || !episode_mini_mask || !returns) return fail(-22, "ic3_returns_scan: null argument");
    if (T <= 0 || E <= 0 || N <= 0 || N > EF_THREADS) return fail(-22, "ic3_returns_scan: bad sizes (1..256 agents per env)");
    const int epb = EF_THREADS / N;
    hipLaunchKernelGGL(returns_scan_kernel, dim3((E + epb - 1) / epb), dim3(EF_THREADS), 0, (hipStream_t)stream, reward, episode_mask,
                       episode_mini_mask, gamma, mean_ratio, returns, T, E, N, epb);
    IC3_HIP(hipGetLastError());
    return 0;
}

size_t ic3_episode_scratch_bytes(int E, int N)
{
    if (E <= 0 || N <= 0 || N > EF_THREADS) return 0;
    const int epb = EF_THREADS / N;
    const size_t nblocks = ((size_t)E + epb - 1) / epb;
    return nblocks * (2 + 2 * (size_t)N) * sizeof(double);
}

int ic3_episode_finalize(const ic3_episode* ep, ic3_stream stream)
{
    Range range("ic3_episode_finalize");
    if (!ep) return fail(-22, "ic3_episode_finalize: null argument");
    if (ep->struct_size != sizeof(ic3_episode))
        return fail(-22, "ic3_episode_finalize: ic3_episode.struct_size is " + std::to_string(ep->struct_size) + ", this library's is " +
                             std::to_string(sizeof(ic3_episode)) + " (built against another ic3_rollout.h?)");
    const ic3_episode& a = *ep;
    if (a.E <= 0 || a.N <= 0 || a.n < 0) return fail(-22, "ic3_episode_finalize: bad sizes");
    if (a.N > EF_THREADS) return fail(-22, "ic3_episode_finalize: more than 256 agents per env");
    if (!a.done || !a.reward || !a.live || !a.alive_mask || !a.episode_mask || !a.episode_mini_mask || !a.live_after ||
        !a.stats || !a.scratch)
        return fail(-22, "ic3_episode_finalize: null buffer");
    const int epb = EF_THREADS / a.N;
    const int nblocks = (a.E + epb - 1) / epb;
    if ((a.alive == nullptr) != (a.is_completed == nullptr))
        return fail(-22, "ic3_episode_finalize: alive and is_completed come together (both or neither)");
    const bool info = a.alive != nullptr, gate = a.gate != nullptr && !a.gate_ones;
    const dim3 grid(nblocks), block(EF_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (info && gate) hipLaunchKernelGGL((episode_finalize_kernel<true, true>), grid, block, 0, st, a, epb);
    else if (info) hipLaunchKernelGGL((episode_finalize_kernel<true, false>), grid, block, 0, st, a, epb);
    else if (gate) hipLaunchKernelGGL((episode_finalize_kernel<false, true>), grid, block, 0, st, a, epb);
    else hipLaunchKernelGGL((episode_finalize_kernel<false, false>), grid, block, 0, st, a, epb);
    IC3_HIP(hipGetLastError());
    hipLaunchKernelGGL(episode_reduce_kernel, dim3(1), dim3(EF_THREADS), 0, (hipStream_t)stream, a, nblocks);
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
