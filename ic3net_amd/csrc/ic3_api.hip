// ic3_api.hip — C ABI entry points of libic3rollout.so (declared in include/ic3_rollout.h).
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>

#include "ic3_common.hpp"
#include "tj_curriculum.hpp"

namespace ic3 {

static thread_local std::string g_err;

namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)();
roctx_push_t g_push = nullptr;
roctx_pop_t g_pop = nullptr;
int g_roctx = -1;   // -1 not probed, 0 off, 1 on

bool roctx_on()
{
    if (g_roctx < 0) {
        g_roctx = 0;
        const char* e = getenv("IC3_ROCTX");
        if (e && atoi(e) != 0) {
            void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
                g_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
                if (g_push && g_pop) g_roctx = 1;
            }
        }
    }
    return g_roctx == 1;
}
}  // namespace

Range::Range(const char* name) : on(roctx_on())
{
    // Every launching entry point starts with a Range: drop whatever error another user of the runtime left in the thread's
    // "last error" slot (e.g. torch reading the return code of a failed hipStreamEndCapture without clearing it), so that the
    // hipGetLastError() check behind OUR launch reports our launch and nothing else.  (The reverse direction — not leaving
    // our own failures behind for the caller's next call — is IC3_HIP's.)  The slot is only cleared when it holds something
    // (peek first), and what is dropped is kept for ic3_last_error()'s reader under IC3_DEBUG_ERRORS.
    if (const hipError_t pending = hipPeekAtLastError(); pending != hipSuccess) {
        static const bool log_dropped = getenv("IC3_DEBUG_ERRORS") != nullptr;
        if (log_dropped) fprintf(stderr, "libic3rollout: %s entered with a pending HIP error of another caller: %s (cleared)\n", name, hipGetErrorString(pending));
        (void)hipGetLastError();
    }
    if (on) g_push(name);
}
Range::~Range()
{
    if (on) g_pop();
}

void set_error(const std::string& msg) { g_err = msg; }

hipError_t ensure_dynamic_lds(const void* func, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> have;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = have[{ func, dev }];
    if (cur >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

// stats: sum over envs of a per-env int32 flag (success / has_failed) + live step counter
__global__ __launch_bounds__(256) void stats_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ tstep,
                                                    const int32_t* __restrict__ acc, double* __restrict__ out, int E,
                                                    int invert)
{
    __shared__ double sh[5][4];
    double v[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };   // flag, t, and the sums over the episodes that ended inside step launches
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int f = flag[e];
        v[0] += (double)(invert ? 1 - f : f);
        v[1] += (double)tstep[e];
        for (int k = 0; k < 3; ++k) v[2 + k] += (double)acc[(size_t)k * E + e];     // (auto-reset): acc[3][E]
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 5; ++k) {
        double x = v[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if (lane == 0) sh[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&out[threadIdx.x], sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

int env_stats(ic3_env* env, ic3_stats* out, hipStream_t s)
{
    IC3_HIP(hipMemsetAsync(env->d_stats, 0, 5 * sizeof(double), s));
    const int E = env->dims.E;
    int blocks = (E + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const bool pp = env->kind == IC3_ENV_PP;
    hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(256), 0, s, env->f(pp ? "success" : "has_failed"), env->f("t"),
                       env->f("acc_success"), env->d_stats, E, pp ? 0 : 1);
    IC3_HIP(hipGetLastError());
    double h[5];
    IC3_HIP(hipMemcpyAsync(h, env->d_stats, sizeof(h), hipMemcpyDeviceToHost, s));
    IC3_HIP(hipStreamSynchronize(s));
    out->success_sum = h[0];
    out->live_env_steps = (int64_t)h[1];
    out->add_rate = pp ? 0.0 : env->add_rate;
    out->episodes = env->resets * (int64_t)E;
    out->auto_success_sum = h[2];
    out->auto_episodes = (int64_t)h[3];
    out->auto_env_steps = (int64_t)h[4];
    return 0;
}

static void add_field(ic3_env* env, const char* name, int64_t count)
{
    int64_t off = env->fields.empty() ? 0 : env->fields.back().off + env->fields.back().count;
    env->fields.push_back({ name, off, count });
}

static int finish_create(ic3_env* env, int device)
{
    env->device = device;
    int64_t words = env->fields.back().off + env->fields.back().count;
    env->dims.state_words = (int32_t)words;
    IC3_HIP(hipSetDevice(device));
    IC3_HIP(hipMalloc(&env->state, (size_t)words * sizeof(int32_t)));
    IC3_HIP(hipMemset(env->state, 0, (size_t)words * sizeof(int32_t)));
    // episode counters start at -1 so that the first reset plays episode 0
    {
        int64_t off = 0, cnt = 0;
        for (auto& f : env->fields)
            if (!strcmp(f.name, "episode")) { off = f.off; cnt = f.count; }
        IC3_HIP(hipMemset(env->state + off, 0xff, (size_t)cnt * sizeof(int32_t)));
    }
    IC3_HIP(hipMalloc(&env->d_err, sizeof(int32_t)));
    IC3_HIP(hipMemset(env->d_err, 0, sizeof(int32_t)));
    IC3_HIP(hipMalloc(&env->d_stats, 5 * sizeof(double)));
    IC3_HIP(hipMalloc(&env->d_thr, sizeof(int32_t)));
    IC3_HIP(hipMemset(env->d_thr, 0, sizeof(int32_t)));
    return 0;
}

}  // namespace ic3

int32_t* ic3_env::f(const char* name) const
{
    for (const auto& fl : fields)
        if (!strcmp(fl.name, name)) return state + fl.off;
    return nullptr;
}

using namespace ic3;

extern "C" {

int ic3_version(void) { return IC3_VERSION; }

int ic3_abi_check(int version, size_t sizeof_policy, size_t sizeof_episode)
{
    if (version != IC3_VERSION)
        return fail(-22, "ic3_abi_check: the caller was built against ic3_rollout.h version " + std::to_string(version) +
                             ", this library is version " + std::to_string(IC3_VERSION));
    if (sizeof_policy != sizeof(ic3_policy) || sizeof_episode != sizeof(ic3_episode))
        return fail(-22, "ic3_abi_check: struct sizes differ (ic3_policy " + std::to_string(sizeof_policy) + " vs " +
                             std::to_string(sizeof(ic3_policy)) + ", ic3_episode " + std::to_string(sizeof_episode) + " vs " +
                             std::to_string(sizeof(ic3_episode)) + ")");
    return 0;
}
const char* ic3_last_error(void) { return g_err.c_str(); }

int ic3_pp_create(const ic3_pp_cfg* cfg, int device, ic3_env** out)
{
    if (!cfg || !out) return fail(-22, "ic3_pp_create: null argument");
    if (cfg->moving_prey) return fail(-38, "moving_prey: NotImplementedError (predator_prey_env.py:84-85)");
    if (cfg->E <= 0 || cfg->N <= 0 || cfg->N + (cfg->enemy_comm ? cfg->nprey : 0) > 64 || cfg->dim <= 0 || cfg->vision < 0)
        return fail(-22, "ic3_pp_create: need E>0, 0<N (+nprey with enemy_comm)<=64, dim>0, vision>=0");
    if (cfg->nprey != 1) return fail(-22, "ic3_pp_create: only nenemies=1 works in the reference (predator_prey_env.py:258)");
    if (cfg->N + cfg->nprey > cfg->dim * cfg->dim) return fail(-22, "ic3_pp_create: more entities than cells");
    if (cfg->mode < 0 || cfg->mode > 2)
        return fail(-22, "Incorrect mode, Available modes: [cooperative|competitive|mixed]");
    ic3_env* env = new (std::nothrow) ic3_env();
    if (!env) return fail(-12, "out of memory");
    env->kind = IC3_ENV_PP;
    env->pp = *cfg;
    ic3_dims& d = env->dims;
    d.kind = IC3_ENV_PP;
    d.E = cfg->E;
    d.N = cfg->N + (cfg->enemy_comm ? cfg->nprey : 0);   // rows seen by the policy (main.py:125-130)
    d.window = 2 * cfg->vision + 1;
    d.vocab = cfg->dim * cfg->dim + 4;               // predator_prey_env.py:103
    d.obs_dim = d.window * d.window * d.vocab;       // :107 (only the product is used, env_wrappers.py:31)
    d.naction = cfg->stay ? 5 : 4;                   // :90-93
    d.grid_h = d.grid_w = cfg->dim;
    const int64_t E = cfg->E, N = cfg->N, T = cfg->N + cfg->nprey;
    add_field(env, "loc_r", E * T);
    add_field(env, "loc_c", E * T);
    add_field(env, "reached", E * N);
    add_field(env, "over", E);
    add_field(env, "success", E);
    add_field(env, "episode", E);
    add_field(env, "t", E);
    for (const char* nm : { "acc_success", "acc_episodes", "acc_steps" }) add_field(env, nm, E);   // auto-reset sums
    int rc = finish_create(env, device);
    if (rc) { ic3_env_destroy(env); return rc; }
    *out = env;
    return 0;
}

int ic3_tj_create(const ic3_tj_cfg* cfg, int device, ic3_env** out)
{
    if (!cfg || !out) return fail(-22, "ic3_tj_create: null argument");
    if (cfg->vocab_type != 0 && cfg->vocab_type != 1) return fail(-22, "vocab_type must be 0 ('bool') or 1 ('scalar')");
    if (cfg->E <= 0 || cfg->N <= 0 || cfg->N > 64 || cfg->vision < 0)
        return fail(-22, "ic3_tj_create: need E>0, 0<N<=64, vision>=0");
    ic3_env* env = new (std::nothrow) ic3_env();
    if (!env) return fail(-12, "out of memory");
    env->kind = IC3_ENV_TJ;
    env->tj = *cfg;
    int h, w, base, npath, narrival, rpa;
    std::string err;
    std::vector<int32_t> road;
    int rc = tj_build_tables(cfg->dim, cfg->vision, cfg->difficulty, &h, &w, &base, &npath, &narrival, &rpa, env->h_grid,
                             env->h_route_off, env->h_route_rc, err, &road);
    if (rc) { delete env; return fail(rc, err); }
    const bool scalar = cfg->vocab_type == 1;
    if (scalar) env->h_grid = road;   // traffic_junction_env.py:301-307: the grid holds OUTSIDE_CLASS 0 / ROAD_CLASS 1
    ic3_dims& d = env->dims;
    d.kind = IC3_ENV_TJ;
    d.E = cfg->E;
    d.N = cfg->N;
    d.window = 2 * cfg->vision + 1;
    // 'bool': vocab_size = BASE + 3 (:134), obs = Tuple(Discrete, Discrete, MultiBinary)            -> 2 + W*W*vocab
    // 'scalar': vocab_size = 2 (:141), obs = Tuple(Discrete, Discrete, MultiDiscrete(dims), MultiBinary) -> 4 + W*W*2
    d.vocab = scalar ? 2 : base + 3;
    d.obs_dim = (scalar ? 4 : 2) + d.window * d.window * d.vocab;   // env_wrappers.py:21-29
    d.naction = 2;                                       // :108
    d.npath = npath;
    d.narrival = narrival;
    d.grid_h = h;
    d.grid_w = w;
    d.max_route_len = 0;
    for (int p = 0; p < npath; ++p) {
        const int len = env->h_route_off[p + 1] - env->h_route_off[p];
        if (len > d.max_route_len) d.max_route_len = len;
    }
    env->exact_rate = env->add_rate = cfg->add_rate_min;  // :103
    env->epoch_last_update = 0;                           // :104
    const int64_t E = cfg->E, N = cfg->N;
    for (const char* nm : { "alive", "wait", "loc_r", "loc_c", "last_act", "route_loc", "route_id", "is_completed" })
        add_field(env, nm, E * N);
    for (const char* nm : { "cars_in_sys", "has_failed", "over", "episode", "t" }) add_field(env, nm, E);
    for (const char* nm : { "acc_success", "acc_episodes", "acc_steps" }) add_field(env, nm, E);   // auto-reset sums
    rc = finish_create(env, device);
    if (rc) { ic3_env_destroy(env); return rc; }
    std::vector<int32_t> packed(env->h_route_rc.size() / 2);
    for (size_t i = 0; i < packed.size(); ++i) packed[i] = (env->h_route_rc[2 * i] << 16) | env->h_route_rc[2 * i + 1];
    hipError_t e1 = hipMalloc(&env->d_grid, env->h_grid.size() * 4);
    hipError_t e2 = hipMalloc(&env->d_route_off, env->h_route_off.size() * 4);
    hipError_t e3 = hipMalloc(&env->d_route_rc, packed.size() * 4);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { ic3_env_destroy(env); return fail(-12, "hipMalloc failed"); }
    {
        // device copy: road ids ('bool'), or road ? 0 : -1 ('scalar') so that one kernel serves both vocabularies
        std::vector<int32_t> dev_grid = env->h_grid;
        if (scalar)
            for (auto& x : dev_grid) x = x ? 0 : -1;
        (void)hipMemcpy(env->d_grid, dev_grid.data(), dev_grid.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(env->d_route_off, env->h_route_off.data(), env->h_route_off.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(env->d_route_rc, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    *out = env;
    return 0;
}

int ic3_env_destroy(ic3_env* env)
{
    if (!env) return 0;
    (void)hipFree(env->state);
    (void)hipFree(env->d_err);
    (void)hipFree(env->d_stats);
    (void)hipFree(env->d_thr);
    (void)hipFree(env->obs_rec);
    (void)hipFree(env->d_grid);
    (void)hipFree(env->d_route_off);
    (void)hipFree(env->d_route_rc);
    delete env;
    return 0;
}

int ic3_env_dims(const ic3_env* env, ic3_dims* out)
{
    if (!env || !out) return fail(-22, "ic3_env_dims: null argument");
    *out = env->dims;
    return 0;
}

int ic3_env_reset(ic3_env* env, int epoch, float* obs, ic3_stream stream)
{
    ic3::Range range_("ic3_env_reset");
    if (!env) return fail(-22, "ic3_env_reset: null handle");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (env->kind == IC3_ENV_PP) {
        rc = pp_reset(env, s);
    } else {
        const ic3_tj_cfg& c = env->tj;   // curriculum: traffic_junction_env.py:196-200,620-626 (tj_curriculum.hpp)
        tj_curriculum_update(c.add_rate_min, c.add_rate_max, c.curr_start, c.curr_end, epoch, env->exact_rate, env->add_rate,
                             env->epoch_last_update);
        rc = tj_reset(env, s);
    }
    if (rc) return rc;
    IC3_HIP(hipMemsetAsync(env->f("acc_success"), 0, (size_t)3 * env->dims.E * sizeof(int32_t), s));   // 3 adjacent fields
    env->resets += 1;
    if (obs) return ic3_env_observe(env, obs, stream);
    return 0;
}

int ic3_env_reset_to(ic3_env* env, int epoch, const int32_t* host_state, size_t bytes, float* obs, ic3_stream stream)
{
    ic3::Range range_("ic3_env_reset_to");
    if (!env) return fail(-22, "ic3_env_reset_to: null handle");
    if (!host_state) return ic3_env_reset(env, epoch, obs, stream);
    if (bytes != (size_t)env->dims.state_words * 4) return fail(-22, "ic3_env_reset_to: size mismatch");
    int rc = ic3_env_reset(env, epoch, nullptr, stream);     // episode bookkeeping, curriculum, accumulators
    if (rc) return rc;
    IC3_HIP(hipMemcpyAsync(env->state, host_state, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    IC3_HIP(hipStreamSynchronize((hipStream_t)stream));      // the host buffer may be reused by the caller
    if (obs) return ic3_env_observe(env, obs, stream);
    return 0;
}

int ic3_env_set_incremental_obs(ic3_env* env, int on)
{
    if (!env) return fail(-22, "ic3_env_set_incremental_obs: null handle");
    env->painted_valid = false;
    env->painted_obs = nullptr;
    if (!on) {
        if (env->obs_rec) (void)hipFree(env->obs_rec);
        env->obs_rec = nullptr;
        return 0;
    }
    if (!env->obs_rec) {
        const int N = env->dims.N, WW = env->dims.window * env->dims.window;
        const size_t words = env->kind == IC3_ENV_PP ? (size_t)2 * N * WW : (size_t)N + (size_t)2 * N * WW;
        IC3_HIP(hipMalloc(&env->obs_rec, (size_t)env->dims.E * words * sizeof(int32_t)));
    }
    return 0;
}

int ic3_env_set_auto_reset(ic3_env* env, int max_steps)
{
    if (!env || max_steps < 0) return fail(-22, "ic3_env_set_auto_reset: bad arguments");
    env->auto_max_steps = max_steps;
    return 0;
}

int ic3_env_observe(ic3_env* env, float* obs, ic3_stream stream)
{
    ic3::Range range_("ic3_env_observe");
    if (!env || !obs) return fail(-22, "ic3_env_observe: null argument");
    env->touch_obs(obs);
    return env->kind == IC3_ENV_PP ? pp_observe(env, obs, (hipStream_t)stream) : tj_observe(env, obs, (hipStream_t)stream);
}

int ic3_env_observe_at(ic3_env* env, const int32_t* snap, float* obs, ic3_stream stream)
{
    ic3::Range range_("ic3_env_observe");
    if (!env || !obs) return fail(-22, "ic3_env_observe_at: null argument");
    env->touch_obs(obs);
    env->view = snap;
    const int rc = env->kind == IC3_ENV_PP ? pp_observe(env, obs, (hipStream_t)stream) : tj_observe(env, obs, (hipStream_t)stream);
    env->view = nullptr;
    return rc;
}

int ic3_env_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table, float* out, int ldo, int H,
                   ic3_stream stream)
{
    ic3::Range range_("ic3_env_encode");
    if (!env || !Wt || !bias || !out) return fail(-22, "ic3_env_encode: null argument");
    if (ldo <= 0) ldo = H;
    if (H <= 0 || (H & 3) || (ldo & 3) || ldo < H) return fail(-22, "ic3_env_encode: H and ldo must be positive multiples of 4");
    return env->kind == IC3_ENV_PP ? pp_encode(env, Wt, bias, loc_table, out, ldo, H, (hipStream_t)stream)
                                   : tj_encode(env, Wt, bias, loc_table, out, ldo, H, (hipStream_t)stream);
}

int ic3_env_encode_at(ic3_env* env, const int32_t* snap, const float* Wt, const float* bias, const float* loc_table, float* out,
                      int ldo, int H, ic3_stream stream)
{
    if (!env) return fail(-22, "ic3_env_encode_at: null handle");
    env->view = snap;          // (null: the live state)
    const int rc = ic3_env_encode(env, Wt, bias, loc_table, out, ldo, H, stream);
    env->view = nullptr;
    return rc;
}

int ic3_env_encode_table(ic3_env* env, const float* Wt, int H, float* loc_table, ic3_stream stream)
{
    if (!env || !Wt || !loc_table) return fail(-22, "ic3_env_encode_table: null argument");
    if (H <= 0 || (H & 3)) return fail(-22, "ic3_env_encode_table: H must be a positive multiple of 4");
    return env->kind == IC3_ENV_PP ? pp_encode_table(env, Wt, H, loc_table, (hipStream_t)stream)
                                   : tj_encode_table(env, Wt, H, loc_table, (hipStream_t)stream);
}

int ic3_env_snapshot(const ic3_env* env, int32_t* snap, ic3_stream stream)
{
    if (!env || !snap) return fail(-22, "ic3_env_snapshot: null argument");
    IC3_HIP(hipMemcpyAsync(snap, env->state, (size_t)env->dims.state_words * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return 0;
}

int64_t ic3_env_encode_backward_work(const ic3_env* env, int H)
{
    if (!env || H <= 0 || (H & 3)) return fail(-22, "ic3_env_encode_backward_work: H must be a positive multiple of 4");
    return env->kind == IC3_ENV_PP ? pp_encode_bwd_work(env, H) : tj_encode_bwd_work(env, H);
}

int ic3_env_encode_backward(ic3_env* env, const int32_t* snap, const float* grad_out, int ldg, int H, float* dWt,
                            float* dbias, float* work, ic3_stream stream)
{
    if (!env || !grad_out || !dWt || !work) return fail(-22, "ic3_env_encode_backward: null argument");
    if (ldg <= 0) ldg = H;
    if (H <= 0 || (H & 3) || (ldg & 3) || ldg < H)
        return fail(-22, "ic3_env_encode_backward: H and ldg must be positive multiples of 4");
    return env->kind == IC3_ENV_PP ? pp_encode_bwd(env, snap, grad_out, ldg, H, dWt, dbias, work, (hipStream_t)stream)
                                   : tj_encode_bwd(env, snap, grad_out, ldg, H, dWt, dbias, work, (hipStream_t)stream);
}

int ic3_env_encode_backward_accumulate(ic3_env* env, const int32_t* snap, const float* grad_out, int ldg, int H, float* work,
                                       int first, ic3_stream stream)
{
    if (!env || !grad_out || !work) return fail(-22, "ic3_env_encode_backward_accumulate: null argument");
    if (ldg <= 0) ldg = H;
    if (H <= 0 || (H & 3) || (ldg & 3) || ldg < H)
        return fail(-22, "ic3_env_encode_backward_accumulate: H and ldg must be positive multiples of 4");
    const int mode = first ? 1 : 2;
    return env->kind == IC3_ENV_PP ? pp_encode_bwd(env, snap, grad_out, ldg, H, nullptr, nullptr, work, (hipStream_t)stream, mode)
                                   : tj_encode_bwd(env, snap, grad_out, ldg, H, nullptr, nullptr, work, (hipStream_t)stream, mode);
}

int ic3_env_encode_backward_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, ic3_stream stream)
{
    if (!env || !dWt || !work) return fail(-22, "ic3_env_encode_backward_finish: null argument");
    if (H <= 0 || (H & 3)) return fail(-22, "ic3_env_encode_backward_finish: H must be a positive multiple of 4");
    return env->kind == IC3_ENV_PP ? pp_encode_bwd(env, nullptr, nullptr, H, H, dWt, dbias, work, (hipStream_t)stream, 3)
                                   : tj_encode_bwd(env, nullptr, nullptr, H, H, dWt, dbias, work, (hipStream_t)stream, 3);
}

int64_t ic3_env_encode_backward_window_work(const ic3_env* env, int H)
{
    if (!env || H <= 0) return fail(-22, "ic3_env_encode_backward_window_work: null handle or hid_size <= 0");
    return env->kind == IC3_ENV_PP ? pp_encode_bwd_window_work(env, H) : tj_encode_bwd_window_work(env, H);
}

int ic3_env_encode_backward_window(ic3_env* env, const int32_t* snaps, int64_t snap_words, int T, const float* grad_out, int ldg,
                                   int64_t step_stride, int H, float* work, int first, ic3_stream stream)
{
    if (!env || !snaps || !grad_out || !work) return fail(-22, "ic3_env_encode_backward_window: null argument");
    if (ldg <= 0) ldg = H;
    if (T <= 0 || H <= 0 || ldg < H || snap_words < env->dims.state_words || step_stride < 0)
        return fail(-22, "ic3_env_encode_backward_window: T > 0, ldg >= H, snap_words >= dims.state_words, step_stride >= 0");
    return env->kind == IC3_ENV_PP
               ? pp_encode_bwd_window(env, snaps, snap_words, T, grad_out, ldg, step_stride, H, work, first, (hipStream_t)stream)
               : tj_encode_bwd_window(env, snaps, snap_words, T, grad_out, ldg, step_stride, H, work, first, (hipStream_t)stream);
}

int ic3_env_encode_backward_window_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, ic3_stream stream)
{
    if (!env || !dWt || !work || H <= 0) return fail(-22, "ic3_env_encode_backward_window_finish: null argument");
    return env->kind == IC3_ENV_PP ? pp_encode_bwd_window_finish(env, H, dWt, dbias, work, (hipStream_t)stream)
                                   : tj_encode_bwd_window_finish(env, H, dWt, dbias, work, (hipStream_t)stream);
}

int ic3_env_step(ic3_env* env, const int32_t* actions, float* obs, float* reward, int32_t* done, int32_t* alive,
                 int32_t* is_completed, ic3_stream stream)
{
    ic3::Range range_("ic3_env_step");
    if (!env || !actions || !reward || !done) return fail(-22, "ic3_env_step: null argument");
    if (env->resets == 0) return fail(-22, "ic3_env_step: reset() has not been called");
    hipStream_t s = (hipStream_t)stream;
    int rc = env->kind == IC3_ENV_PP ? pp_step(env, actions, reward, done, alive, is_completed, s)
                                     : tj_step(env, actions, reward, done, alive, is_completed, s);
    if (rc) return rc;
    if (obs) return ic3_env_observe(env, obs, stream);
    return 0;
}

int ic3_env_sample_actions(const ic3_env* env, const float* logp, int ld, int A, int head, int32_t* action,
                           float* chosen_logp, ic3_stream stream)
{
    ic3::Range range_("ic3_env_sample_actions");
    if (!env || !logp || !action || A <= 0) return fail(-22, "ic3_env_sample_actions: bad arguments");
    return sample_actions_env(env, logp, ld, A, head, action, chosen_logp, (hipStream_t)stream);
}

int ic3_env_check(ic3_env* env, ic3_stream stream)
{
    if (!env) return fail(-22, "ic3_env_check: null handle");
    int32_t h = 0;
    IC3_HIP(hipMemcpyAsync(&h, env->d_err, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    IC3_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (h) {
        IC3_HIP(hipMemsetAsync(env->d_err, 0, sizeof(int32_t), (hipStream_t)stream));
        return fail(-22, "Actions should be in the range [0,naction).");
    }
    return 0;
}

int ic3_env_get_state(const ic3_env* env, int32_t* host_out, size_t bytes, ic3_stream stream)
{
    if (!env || !host_out) return fail(-22, "ic3_env_get_state: null argument");
    if (bytes != (size_t)env->dims.state_words * 4) return fail(-22, "ic3_env_get_state: size mismatch");
    IC3_HIP(hipMemcpyAsync(host_out, env->state, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    IC3_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int ic3_env_set_state(ic3_env* env, const int32_t* host_in, size_t bytes, ic3_stream stream)
{
    if (!env || !host_in) return fail(-22, "ic3_env_set_state: null argument");
    if (bytes != (size_t)env->dims.state_words * 4) return fail(-22, "ic3_env_set_state: size mismatch");
    IC3_HIP(hipMemcpyAsync(env->state, host_in, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    IC3_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (env->resets == 0) env->resets = 1;
    return 0;
}

int ic3_env_state_field(const ic3_env* env, const char* name, int64_t* offset_words, int64_t* count_words)
{
    if (!env || !name) return fail(-22, "ic3_env_state_field: null argument");
    for (const auto& f : env->fields)
        if (!strcmp(f.name, name)) {
            if (offset_words) *offset_words = f.off;
            if (count_words) *count_words = f.count;
            return 0;
        }
    return fail(-2, std::string("no such state field: ") + name);
}

int ic3_tj_get_tables(const ic3_env* env, int32_t* grid, int32_t* route_off, int32_t* route_rc, size_t rc_capacity_words)
{
    if (!env || env->kind != IC3_ENV_TJ) return fail(-22, "ic3_tj_get_tables: not a Traffic-Junction handle");
    if (grid) memcpy(grid, env->h_grid.data(), env->h_grid.size() * 4);
    if (route_off) memcpy(route_off, env->h_route_off.data(), env->h_route_off.size() * 4);
    if (route_rc) {
        if (rc_capacity_words < env->h_route_rc.size()) return fail(-22, "ic3_tj_get_tables: route_rc buffer too small");
        memcpy(route_rc, env->h_route_rc.data(), env->h_route_rc.size() * 4);
    }
    return (int)env->h_route_rc.size();  // words needed for route_rc
}

int ic3_tj_build_tables(int dim, int vision, int difficulty, ic3_dims* dims_out, int32_t* grid, int32_t* route_off,
                        int32_t* route_rc, size_t rc_capacity_words)
{
    int h, w, base, npath, narrival, rpa;
    std::vector<int32_t> g, off, rc;
    std::string err;
    int r = tj_build_tables(dim, vision, difficulty, &h, &w, &base, &npath, &narrival, &rpa, g, off, rc, err);
    if (r) return fail(r, err);
    if (dims_out) {
        memset(dims_out, 0, sizeof(*dims_out));
        dims_out->kind = IC3_ENV_TJ;
        dims_out->window = 2 * vision + 1;
        dims_out->vocab = base + 3;
        dims_out->obs_dim = 2 + dims_out->window * dims_out->window * dims_out->vocab;
        dims_out->naction = 2;
        dims_out->npath = npath;
        dims_out->narrival = narrival;
        dims_out->grid_h = h;
        dims_out->grid_w = w;
        for (int p = 0; p < npath; ++p)
            if (off[p + 1] - off[p] > dims_out->max_route_len) dims_out->max_route_len = off[p + 1] - off[p];
    }
    if (grid) memcpy(grid, g.data(), g.size() * 4);
    if (route_off) memcpy(route_off, off.data(), off.size() * 4);
    if (route_rc) {
        if (rc_capacity_words < rc.size()) return fail(-22, "ic3_tj_build_tables: route_rc buffer too small");
        memcpy(route_rc, rc.data(), rc.size() * 4);
    }
    return (int)rc.size();
}

int ic3_tj_get_add_rate(const ic3_env* env, double* add_rate, double* exact_rate)
{
    if (!env || env->kind != IC3_ENV_TJ) return fail(-22, "ic3_tj_get_add_rate: not a Traffic-Junction handle");
    if (add_rate) *add_rate = env->add_rate;
    if (exact_rate) *exact_rate = env->exact_rate;
    return 0;
}

int ic3_event_create(void** event)
{
    if (!event) return fail(-22, "ic3_event_create: null argument");
    hipEvent_t e;
    IC3_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return 0;
}

int ic3_event_destroy(void* event)
{
    if (event) IC3_HIP(hipEventDestroy((hipEvent_t)event));
    return 0;
}

int ic3_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!start || !stop || !ms) return fail(-22, "ic3_event_elapsed_ms: null argument");
    IC3_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int ic3_env_set_step_events(ic3_env* env, void* start, void* stop)
{
    if (!env) return fail(-22, "ic3_env_set_step_events: null argument");
    env->ev_start = start;
    env->ev_stop = stop;
    return 0;
}

int ic3_env_stats(ic3_env* env, ic3_stats* host_out, ic3_stream stream)
{
    if (!env || !host_out) return fail(-22, "ic3_env_stats: null argument");
    return env_stats(env, host_out, (hipStream_t)stream);
}

}  // extern "C"
