// ic3_common.hpp — shared declarations of libic3rollout (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "ic3_rollout.h"

// The dynamic LDS of a kernel has one spelling, so that tests/host (the same sources behind the same C ABI on a CPU, under
// ASan / UBSan) can give it storage.
#ifndef IC3_DYNAMIC_LDS
#define IC3_DYNAMIC_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif
// Likewise the compiler fences of the hand-scheduled kernels: the value is opaque to the optimiser from here on and lives
// in a scalar / vector register; wait for every outstanding vector-memory operation of the wave.
#ifndef IC3_OPAQUE_SGPR
#define IC3_OPAQUE_SGPR(x) asm volatile("" : "+s"(x))
#define IC3_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#define IC3_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define IC3_WAIT_VMEM_N(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")   // at most n vector-memory operations in flight
#endif

namespace ic3 {

// Random stream contract: x24 = Philox4x32-10((draw, t, episode, domain), (seed, env_gid))[0] >> 8
enum : uint32_t { DOMAIN_PP_RESET = 1, DOMAIN_TJ_ADD = 2, DOMAIN_SAMPLE = 3, DOMAIN_BENCH = 4 };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ inline uint32_t philox_x24(uint32_t seed, uint32_t env_gid, uint32_t domain, uint32_t episode,
                                               uint32_t t, uint32_t draw)
{
    uint32_t c0 = draw, c1 = t, c2 = episode, c3 = domain, k0 = seed, k1 = env_gid;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0 >> 8;
}

// floor(u * n) with u = x24 / 2^24, exact in integers
__host__ __device__ inline uint32_t scale24(uint32_t x24, uint32_t n) { return (uint32_t)(((uint64_t)x24 * n) >> 24); }

struct Field {
    const char* name;
    int64_t off;    // int32 words from state base
    int64_t count;  // int32 words
};

// smallest power of two >= n (n <= 64): lanes per environment in the per-agent kernels
inline int group_lanes(int n)
{
    int g = 1;
    while (g < n) g <<= 1;
    return g;
}

}  // namespace ic3

struct ic3_env {
    int kind = 0;
    int device = 0;
    ic3_dims dims{};
    ic3_pp_cfg pp{};
    ic3_tj_cfg tj{};
    int32_t* state = nullptr;  // device, dims.state_words int32
    std::vector<ic3::Field> fields;
    int32_t* d_err = nullptr;  // sticky bad-action flag
    double* d_stats = nullptr; // small device scratch for ic3_env_stats
    int32_t* d_thr = nullptr;  // TJ: floor(add_rate * 2^24), device-resident so captured step graphs stay valid
    int64_t resets = 0;
    int auto_max_steps = 0;    // ic3_env_set_auto_reset: > 0 = finished envs restart inside the step launch
    void *ev_start = nullptr, *ev_stop = nullptr;   // ic3_env_set_step_events: recorded by the next ic3_policy_step launch
    float *h_out = nullptr, *c_out = nullptr;       // ic3_env_set_hidden_out: where the next ic3_policy_step writes h', c'
    float *gates_out = nullptr, *xh_out = nullptr;  // ic3_env_set_record_out: where the next ic3_policy_step records its cell's gates / inp rows
    // ic3_env_set_incremental_obs (opt-in experiment): per-env record of what ic3_policy_step painted into `painted_obs`
    int32_t* obs_rec = nullptr;
    const float* painted_obs = nullptr;
    bool painted_valid = false;
    void touch_obs(const float* obs)        // another writer of that buffer
    {
        if (obs && obs == painted_obs) painted_valid = false;
    }
    // Traffic-Junction constant tables (device + host copies)
    int32_t* d_grid = nullptr;       // [h*w] road ids
    int32_t* d_route_off = nullptr;  // [npath+1]
    int32_t* d_route_rc = nullptr;   // [total] packed (row << 16 | col)
    std::vector<int32_t> h_grid, h_route_off, h_route_rc;  // h_route_rc: (row, col) pairs
    // curriculum scalars (traffic_junction_env.py:103-104)
    double exact_rate = 0, add_rate = 0, epoch_last_update = 0;

    int32_t* f(const char* name) const;
    const int32_t* view = nullptr;  // when set: the observation kernels read this snapshot instead of `state`
    const int32_t* fv(const char* name) const { return (view ? view : state) + (f(name) - state); }
};

namespace ic3 {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// hipFuncAttributeMaxDynamicSharedMemorySize of `func` raised to >= `bytes` on the CURRENT device.  The attribute is per
// device; what has been set is remembered per (function, device), so the driver call happens once per pair, not per launch.
hipError_t ensure_dynamic_lds(const void* func, size_t bytes);

// roctx ranges around the launches of the hot loop (the reference's unused utils.Timer, utils.py:86-98):
// `IC3_ROCTX=1 rocprofv3 --marker-trace --kernel-trace ...` shows reset / step / observe / encode / policy_step /
// sample spans above the kernels.  librocprofiler-sdk-roctx is dlopen()ed on first use; without IC3_ROCTX the
// constructor is one predictable branch.
struct Range {
    explicit Range(const char* name);
    ~Range();
    bool on;
};

// A failed runtime call is reported through the return code + ic3_last_error(); the runtime's own "last error" is read
// back so that the caller's next HIP call (torch's, say) does not trip over it.
#define IC3_HIP(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            (void)hipGetLastError();                                                                      \
            return ic3::fail(-5 /*EIO*/, std::string(#expr) + ": " + hipGetErrorString(_e));               \
        }                                                                                                 \
    } while (0)

// LSTM nonlinearities on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each), shared by the
// pointwise cell kernels and the fused MFMA kernel.  The accurate ocml expf/tanhf made those kernels VALU-bound
// (~4000 VALU cycles per wave against ~700 cycles of memory time in lstm_cell_heads_kernel).  Absolute error <= ~2e-7
// on outputs in [-1, 1] (the parity bar is 1e-5); overflow-safe: exp2(+big) = inf -> 1/(1+inf) = 0.
#if defined(__HIPCC__)
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x)
{
    // (an explicit fma: which of mul + sub / fma the compiler picks must not depend on the code around the call)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.0f);
}
#endif

// The communication block's masked sums (comm.py:181-205 in closed form: S = sum_i m_i h_i, out_j = m_j (S - m_j h_j) scale) on
// ONE-COMPONENT instructions.  Written on float4 values (`S += m * x`) hipcc packs the arithmetic into v_pk_fma_f32 / v_pk_mul_f32
// with op_sel broadcasting the mask out of the destination pair of a ds_read2_b32 — and on gfx950 that sequence came out WRONG
// now and then: lanes 48..63 of the wave took a stale multiplier in the low half of one packed FMA, i.e. one agent's row was
// missing from (or doubled in) S for the hidden columns >= 64 of the second env of a wave (round 6: ~1 % of the TJ-medium
// E = 8192 launches of policy_step_kernel<128, TJ, split>, one env per launch off by ~1e-3 — found through a test that failed one
// suite run in five; tools/exp/flake_probe2.py, profiles/r06/packed_fma_hazard.txt).  The same loop on v_fma_f32, one component
// at a time behind opaque-register fences so that the SLP vectoriser cannot re-pack it: 0 differing runs in 40 (packed: 40 in 40).
// Same rounding as the packed form (a fused multiply-add per term).
#if defined(__HIPCC__)
typedef float ic3_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ic3_f32x4 mask_fma4(float m, ic3_f32x4 x, ic3_f32x4 s)          // s + m x
{
    float s0 = __builtin_fmaf(m, x[0], s[0]);
    IC3_OPAQUE_VGPR(s0);
    float s1 = __builtin_fmaf(m, x[1], s[1]);
    IC3_OPAQUE_VGPR(s1);
    float s2 = __builtin_fmaf(m, x[2], s[2]);
    IC3_OPAQUE_VGPR(s2);
    float s3 = __builtin_fmaf(m, x[3], s[3]);
    IC3_OPAQUE_VGPR(s3);
    return ic3_f32x4{ s0, s1, s2, s3 };
}
__device__ __forceinline__ ic3_f32x4 comm_out4(float m, ic3_f32x4 S, ic3_f32x4 x, float scale)   // m (S - m x) scale
{
    float o0 = (m * __builtin_fmaf(-m, x[0], S[0])) * scale;
    IC3_OPAQUE_VGPR(o0);
    float o1 = (m * __builtin_fmaf(-m, x[1], S[1])) * scale;
    IC3_OPAQUE_VGPR(o1);
    float o2 = (m * __builtin_fmaf(-m, x[2], S[2])) * scale;
    IC3_OPAQUE_VGPR(o2);
    float o3 = (m * __builtin_fmaf(-m, x[3], S[3])) * scale;
    IC3_OPAQUE_VGPR(o3);
    return ic3_f32x4{ o0, o1, o2, o3 };
}
#endif

// pp_kernels.hip
int pp_reset(ic3_env* env, hipStream_t s);
int pp_step(ic3_env* env, const int32_t* actions, float* reward, int32_t* done, int32_t* alive, int32_t* is_completed,
            hipStream_t s);
int pp_observe(ic3_env* env, float* obs, hipStream_t s);
int pp_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table, float* out, int ldo, int H,
              hipStream_t s);
int pp_encode_table(ic3_env* env, const float* Wt, int H, float* table, hipStream_t s);
// tj_kernels.hip
int tj_reset(ic3_env* env, hipStream_t s);
int tj_step(ic3_env* env, const int32_t* actions, float* reward, int32_t* done, int32_t* alive, int32_t* is_completed,
            hipStream_t s);
int tj_observe(ic3_env* env, float* obs, hipStream_t s);
int tj_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table, float* out, int ldo, int H,
              hipStream_t s);
int tj_encode_table(ic3_env* env, const float* Wt, int H, float* table, hipStream_t s);
// sparse-encoder backward (enc_bwd.hpp)
int64_t pp_encode_bwd_work(const ic3_env* env, int H);
int64_t tj_encode_bwd_work(const ic3_env* env, int H);
// mode 0: both stages; 1 / 2: stage 1 writing / adding to the partials in `work`; 3: stage 2 alone
int pp_encode_bwd(ic3_env* env, const int32_t* snap, const float* g, int ldg, int H, float* dWt, float* dbias, float* work,
                  hipStream_t s, int mode = 0);
int tj_encode_bwd(ic3_env* env, const int32_t* snap, const float* g, int ldg, int H, float* dWt, float* dbias, float* work,
                  hipStream_t s, int mode = 0);
// window form: stage 1 over T recorded states in one launch, and its expand stage
int enc_bwd_cus();
int64_t pp_encode_bwd_window_work(const ic3_env* env, int H);
int64_t tj_encode_bwd_window_work(const ic3_env* env, int H);
int pp_encode_bwd_window(ic3_env* env, const int32_t* snaps, long long snap_words, int T, const float* g, int ldg, long long g_step,
                         int H, float* work, int first, hipStream_t s);
int tj_encode_bwd_window(ic3_env* env, const int32_t* snaps, long long snap_words, int T, const float* g, int ldg, long long g_step,
                         int H, float* work, int first, hipStream_t s);
int pp_encode_bwd_window_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, hipStream_t s);
int tj_encode_bwd_window_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, hipStream_t s);
// tj_tables.cpp (host)
int tj_build_tables(int dim, int vision, int difficulty, int* h, int* w, int* base, int* npath, int* narrival,
                    int* routes_per_arrival, std::vector<int32_t>& grid, std::vector<int32_t>& route_off,
                    std::vector<int32_t>& route_rc, std::string& err, std::vector<int32_t>* road_out = nullptr);
// policy_ops.hip
int sample_actions_env(const ic3_env* env, const float* logp, int ld, int A, int head, int32_t* action, float* chosen_logp,
                       hipStream_t s);
// stats
int env_stats(ic3_env* env, ic3_stats* out, hipStream_t s);

}  // namespace ic3
