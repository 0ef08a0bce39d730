// ic3_episode_finalize — what get_episode derives per step next to the policy / env calls
// (/root/reference/trainer.py:70-105,109-110): the live / alive / episode masks of every transition and the
// per-agent reward and comm-action sums, for n lock-step slots of E envs, as ONE launch over the episode buffers the
// step launches wrote ([T][E] / [T][E][N], step-major).  One thread per (env, agent) walks the n slots (its env's
// `done` history is a running product), so every access is coalesced across the (env, agent) index; the sums are
// reduced in a fixed order (block partials, then the last block to finish), so the statistics are reproducible.
#include <hip/hip_runtime.h>

#include "ic3_common.hpp"

namespace ic3 {

constexpr int EF_THREADS = 256;

// stats layout: [0] num_steps (sum of live), [1] envs whose last held slot has done set, [2, 2+N) reward sums,
// [2+N, 2+2N) gate (comm_action) sums
__global__ __launch_bounds__(EF_THREADS) void episode_finalize_kernel(ic3_episode a, int epb /* envs per block */, int nblocks)
{
    __shared__ double sh[EF_THREADS][2];
    __shared__ double shn[EF_THREADS][2];
    __shared__ int last_block;
    const int tid = threadIdx.x, N = a.N, E = a.E, n = a.n;
    const int el = tid / N, ag = tid - el * N;
    const int e = blockIdx.x * epb + el;
    const bool valid = el < epb && e < E;
    double rsum = 0.0, gsum = 0.0, steps = 0.0, zero_len = 0.0;
    if (valid) {
        const size_t EN = (size_t)E * N;
        const size_t en = (size_t)e * N + ag;
        float live = 1.0f;
        for (int t = 0; t < n; ++t) {
            const bool d = a.done[(size_t)t * E + e] != 0;
            const bool done_t = d || (a.forced_last && t == n - 1);            // trainer.py:90
            const size_t i = (size_t)t * EN + en;
            const float al = a.alive ? (float)a.alive[i] : 1.0f;                // trainer.py:78-81
            a.alive_mask[i] = al * live;
            float mini = 1.0f;                                                  // trainer.py:98-99 (only when not done, Q26)
            if (a.is_completed && !done_t) mini = 1.0f - (float)a.is_completed[i];
            a.episode_mini_mask[i] = mini;
            rsum += (double)a.reward[i];                                        // trainer.py:86
            if (a.gate_ones) gsum += (double)live;                              // trainer.py:73-75
            else if (a.gate) gsum += (double)((float)a.gate[(size_t)t * a.gate_stride + en] * live);
            if (ag == 0) {
                a.live[(size_t)t * E + e] = live;
                a.episode_mask[(size_t)t * E + e] = done_t ? 0.0f : 1.0f;        // trainer.py:92-96
                steps += (double)live;                                          // trainer.py:109-110
            }
            if (!a.auto_reset && d) live = 0.0f;     // lock-step: a finished env idles (auto-reset: every slot is real)
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
    __threadfence();
    __syncthreads();
    if (tid == 0) last_block = (atomicAdd(a.counter, 1) == nblocks - 1);
    __syncthreads();
    if (!last_block) return;
    __threadfence();
    for (int k = tid; k < 2 + 2 * N; k += EF_THREADS) {      // fixed order over the blocks
        double v = 0.0;
        for (int b = 0; b < nblocks; ++b) v += __builtin_nontemporal_load(a.scratch + (size_t)b * (2 + 2 * N) + k);
        a.stats[k] = v;
    }
    if (tid == 0) *a.counter = 0;                            // ready for the next call
}

}  // namespace ic3

using namespace ic3;

extern "C" {

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
    const ic3_episode& a = *ep;
    if (a.E <= 0 || a.N <= 0 || a.n < 0) return fail(-22, "ic3_episode_finalize: bad sizes");
    if (a.N > EF_THREADS) return fail(-22, "ic3_episode_finalize: more than 256 agents per env");
    if (!a.done || !a.reward || !a.live || !a.alive_mask || !a.episode_mask || !a.episode_mini_mask || !a.live_after ||
        !a.stats || !a.scratch || !a.counter)
        return fail(-22, "ic3_episode_finalize: null buffer");
    const int epb = EF_THREADS / a.N;
    const int nblocks = (a.E + epb - 1) / epb;
    hipLaunchKernelGGL(episode_finalize_kernel, dim3(nblocks), dim3(EF_THREADS), 0, (hipStream_t)stream, a, epb, nblocks);
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
