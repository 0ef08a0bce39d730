// policy_ops.hip — the two custom policy-side ops of the rollout step (gfx950):
//   ic3_comm_masked_mean : CommNetMLP communication block, /root/reference/comm.py:181-205
//   ic3_sample_actions   : select_action, /root/reference/action_utils.py:32-36
// plus synthetic uniform actions for env-only benchmarks.
#include "ic3_common.hpp"

namespace ic3 {

// comm.py:181-205 builds an (B,N,N,H) expand of h, multiplies by (1-eye), 1/(n_alive-1), the sender
// mask and the receiver mask, and sums over senders: O(N^2 H) memory traffic in fp64.  Closed form
// per env (SURVEY B.5 i, probed exact to 2e-16 against the reference):
//     m_j  = alive_j * comm_action_j
//     out_j = m_j * (S - m_j * h_j) * scale,   S = sum_i m_i h_i,   scale = 1/(n_alive-1) if avg && n_alive>1
// One workgroup per env, one lane per hidden column: h rows are read coalesced (and re-read from L1/L2
// for the second pass), 2*N*H*4 algorithmic bytes per env.
__global__ __launch_bounds__(256) void comm_masked_mean_kernel(const float* __restrict__ h,
                                                               const int32_t* __restrict__ alive,
                                                               const int32_t* __restrict__ comm_action,
                                                               float* __restrict__ out, int N, int H, int mode_avg,
                                                               int mask_self)
{
    const int e = blockIdx.x;
    const float* he = h + (size_t)e * N * H;
    float* oe = out + (size_t)e * N * H;
    int n_alive = 0;
    for (int j = 0; j < N; ++j) n_alive += alive ? alive[(size_t)e * N + j] : 1;  // comm.py:102-107, quirk Q21
    const float scale = (mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;  // comm.py:194-196, Q23
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        if (!mask_self) {  // comm_mask_zero: comm.py:40-41 -> all-zero communication
            for (int j = 0; j < N; ++j) oe[(size_t)j * H + k] = 0.0f;
            continue;
        }
        float S = 0.0f;
        for (int i = 0; i < N; ++i) {
            const int m = (alive ? alive[(size_t)e * N + i] : 1) * (comm_action ? comm_action[(size_t)e * N + i] : 1);
            S += (float)m * he[(size_t)i * H + k];
        }
        for (int j = 0; j < N; ++j) {
            const float m = (float)((alive ? alive[(size_t)e * N + j] : 1) *
                                    (comm_action ? comm_action[(size_t)e * N + j] : 1));
            oe[(size_t)j * H + k] = m * (S - m * he[(size_t)j * H + k]) * scale;
        }
    }
}

// action_utils.py:32-36: torch.multinomial(exp(logp), 1) per row.  Inverse-CDF on the injected uniform:
// first a with u < sum_{b<=a} exp(logp_b), last action as fallback (fp32, left-to-right).
__global__ __launch_bounds__(256) void sample_actions_kernel(const float* __restrict__ logp, int A, int head,
                                                             uint32_t seed, uint32_t gid0, uint32_t episode, uint32_t t,
                                                             int32_t* __restrict__ action,
                                                             float* __restrict__ chosen_logp, int E, int N)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= E * N) return;
    const int e = row / N, n = row - e * N;
    const uint32_t x = philox_x24(seed, gid0 + (uint32_t)e, DOMAIN_SAMPLE, episode, t, (uint32_t)(head * N + n));
    const float u = (float)x * (1.0f / 16777216.0f);
    const float* lp = logp + (size_t)row * A;
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

extern "C" int ic3_comm_masked_mean(const float* h, const int32_t* alive, const int32_t* comm_action, float* out, int E,
                                    int N, int H, int mode_avg, int mask_self, ic3_stream stream)
{
    if (!h || !out || E <= 0 || N <= 0 || H <= 0) return ic3::fail(-22, "ic3_comm_masked_mean: bad arguments");
    const int threads = H >= 256 ? 256 : ((H + 63) / 64) * 64;
    hipLaunchKernelGGL(ic3::comm_masked_mean_kernel, dim3(E), dim3(threads), 0, (hipStream_t)stream, h, alive,
                       comm_action, out, N, H, mode_avg, mask_self);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_sample_actions(const float* logp, int A, int head, uint32_t seed, uint32_t env_id_offset,
                                  uint32_t episode, uint32_t t, int32_t* action, float* chosen_logp, int E, int N,
                                  ic3_stream stream)
{
    if (!logp || !action || A <= 0 || E <= 0 || N <= 0) return ic3::fail(-22, "ic3_sample_actions: bad arguments");
    const int rows = E * N;
    hipLaunchKernelGGL(ic3::sample_actions_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, logp, A,
                       head, seed, env_id_offset, episode, t, action, chosen_logp, E, N);
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
