// tj_kernels.hip — Traffic-Junction: reset / step / observation assembly for E environments (gfx950).
//
// Reference semantics: /root/reference/ic3net-envs/ic3net_envs/traffic_junction_env.py (cited "TJ:line").
// One lane group (G = pow2 >= max(N, 8) lanes, inside one wavefront) per environment; the order-dependent
// `_add_cars` loop runs as <= 8 uniform iterations with ballot-ranked dead-slot selection, the O(N^2)
// collision test as N lane broadcasts.
#include <algorithm>

#include "enc_bwd.hpp"
#include "env_device.hpp"
#include "tj_curriculum.hpp"
#include "ic3_common.hpp"

namespace ic3 {

__global__ __launch_bounds__(256) void tj_reset_kernel(int32_t* __restrict__ alive, int32_t* __restrict__ wait,
                                                       int32_t* __restrict__ loc_r, int32_t* __restrict__ loc_c,
                                                       int32_t* __restrict__ last_act, int32_t* __restrict__ route_loc,
                                                       int32_t* __restrict__ route_id, int32_t* __restrict__ completed,
                                                       int32_t* __restrict__ cars, int32_t* __restrict__ failed,
                                                       int32_t* __restrict__ over, int32_t* __restrict__ episode,
                                                       int32_t* __restrict__ tstep, int E, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E * N) tj_reset_car(alive, wait, loc_r, loc_c, last_act, route_loc, route_id, completed, (size_t)i);
    if (i < E) {
        cars[i] = 0;    // TJ:173
        failed[i] = 0;  // TJ:169
        over[i] = 0;    // TJ:168
        episode[i] += 1;
        tstep[i] = 0;
    }
}

// The step body lives in env_device.hpp (tj_step_lanes): the fused policy+step kernel runs the same code.
__global__ __launch_bounds__(256) void tj_step_kernel(TJState st, StepOut out, const int32_t* __restrict__ actions, int E,
                                                      int G)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / G, n = tid - e * G;
    tj_step_lanes(st, out, e, n, E, G, [&]() { return actions[(size_t)e * st.N + n]; });
}

// TJ:321-366 _get_obs ('bool' vocab) + env_wrappers.py:88-100: row a = [last_act/(naction-1),
// route_id/(npath-1), one-hot window], all-zero if the car is dead.  CAR channel counts every car on
// the cell including dead ones parked at (0,0) (quirk Q8).  Rows are 2+W*W*vocab floats (not 16-byte
// multiples), so this path uses coalesced dword stores; algorithmic bytes per env = N*obs_dim*4.
__global__ __launch_bounds__(256) void tj_obs_kernel(const int32_t* __restrict__ alive_s,
                                                     const int32_t* __restrict__ loc_r, const int32_t* __restrict__ loc_c,
                                                     const int32_t* __restrict__ last_act_s,
                                                     const int32_t* __restrict__ route_id_s,
                                                     const int32_t* __restrict__ grid, float* __restrict__ obs, int N,
                                                     int h, int w, int v, int vocab, int outside, int car_class, int npath,
                                                     int hdr)
{
    // hdr = 2 ('bool' vocab: [last_act, route]) or 4 ('scalar' vocab: + p_norm = (r/(h-1), c/(w-1)), TJ:344,361).
    // In scalar mode the uploaded grid is (road ? 0 : -1), vocab = 2, car_class = 1, outside = -1, so the one-hot
    // formula below yields exactly the reference's (road, #cars) pair per window cell (TJ:331-332).
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x;
    const int W = 2 * v + 1, WW = W * W, nseg = N * WW, obs_dim = hdr + WW * vocab;
    int32_t* sr = smem;          // [N]
    int32_t* sc = sr + N;        // [N]
    int32_t* sal = sc + N;       // [N]
    float* s0 = reinterpret_cast<float*>(sal + N);  // [N] last_act scalar
    float* s1 = s0 + N;                             // [N] route scalar
    float* s2 = s1 + N;                             // [N] r / (h-1)
    float* s3 = s2 + N;                             // [N] c / (w-1)
    int2* tab = reinterpret_cast<int2*>(smem + ((7 * N + 3) & ~3));  // [nseg] (one-hot channel, #cars)
    for (int a = threadIdx.x; a < N; a += blockDim.x) {
        const size_t i = (size_t)e * N + a;
        sr[a] = loc_r[i];
        sc[a] = loc_c[i];
        sal[a] = alive_s[i];
        s0[a] = (float)((double)last_act_s[i] / 1.0);                       // TJ:338 naction-1 == 1
        s1[a] = (float)((double)route_id_s[i] / (double)(npath - 1));       // TJ:341
        s2[a] = (float)((double)sr[a] / (double)(h - 1));                   // TJ:344
        s3[a] = (float)((double)sc[a] / (double)(w - 1));
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nseg; s += blockDim.x) {
        const int a = s / WW, q = s - a * WW;
        const int dy = q / W, dx = q - dy * W;
        const int gr = sr[a] + dy - v, gc = sc[a] + dx - v;
        const int id = (gr >= 0 && gr < h && gc >= 0 && gc < w) ? grid[gr * w + gc] : outside;  // pad_grid TJ:317
        int ncar = 0;
        for (int p = 0; p < N; ++p) ncar += (sr[p] == gr) & (sc[p] == gc);                    // TJ:326-327
        tab[s] = make_int2(id, ncar);
    }
    __syncthreads();
    const int total = N * obs_dim;
    float* out = obs + (size_t)e * total;
    const float inv_vocab = 1.0f / (float)vocab;
    const int NT = blockDim.x;
    int a = threadIdx.x / obs_dim, off = threadIdx.x - a * obs_dim;
    const int da = NT / obs_dim, doff = NT - da * obs_dim;
    for (int g = threadIdx.x; g < total; g += NT) {
        float z = 0.0f;
        if (sal[a]) {  // TJ:352-356
            if (off == 0) z = s0[a];
            else if (off == 1) z = s1[a];
            else if (off < hdr) z = (off == 2) ? s2[a] : s3[a];
            else {
                const int k = off - hdr;
                const int seg = (int)(((float)k + 0.5f) * inv_vocab);  // exact for k < 2^20
                const int ch = k - seg * vocab;
                const int2 t = tab[a * WW + seg];
                z = (ch == t.x) ? 1.0f : 0.0f;
                if (ch == car_class) z += (float)t.y;
            }
        }
        out[g] = z;
        a += da;
        off += doff;
        if (off >= obs_dim) {
            off -= obs_dim;
            ++a;
        }
    }
}

// float4 form of tj_obs_kernel for large rows: an env chunk (N*obs_dim floats) is in general neither 16-byte aligned
// nor a multiple of 16 bytes, so the workgroup writes <= 3 head and <= 3 tail floats with dword stores and the body
// as float4s whose lane mapping is shifted so that every wave store is 1 KiB-aligned in the global address space;
// each of a float4's four elements is evaluated on its own (it may sit in another row / window cell).
__global__ __launch_bounds__(1024) void tj_obs_vec4_kernel(const int32_t* __restrict__ alive_s,
                                                           const int32_t* __restrict__ loc_r,
                                                           const int32_t* __restrict__ loc_c,
                                                           const int32_t* __restrict__ last_act_s,
                                                           const int32_t* __restrict__ route_id_s,
                                                           const int32_t* __restrict__ grid, float* __restrict__ obs,
                                                           int N, int h, int w, int v, int vocab, int outside,
                                                           int car_class, int npath, int hdr)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x, NT = blockDim.x;
    const int W = 2 * v + 1, WW = W * W, nseg = N * WW, obs_dim = hdr + WW * vocab;
    int32_t* sr = smem;
    int32_t* sc = sr + N;
    int32_t* sal = sc + N;
    float* s0 = reinterpret_cast<float*>(sal + N);
    float* s1 = s0 + N;
    float* s2 = s1 + N;
    float* s3 = s2 + N;
    int2* tab = reinterpret_cast<int2*>(smem + ((7 * N + 3) & ~3));
    for (int a = threadIdx.x; a < N; a += NT) {
        const size_t i = (size_t)e * N + a;
        sr[a] = loc_r[i];
        sc[a] = loc_c[i];
        sal[a] = alive_s[i];
        s0[a] = (float)((double)last_act_s[i] / 1.0);
        s1[a] = (float)((double)route_id_s[i] / (double)(npath - 1));
        s2[a] = (float)((double)sr[a] / (double)(h - 1));
        s3[a] = (float)((double)sc[a] / (double)(w - 1));
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nseg; s += NT) {
        const int a = s / WW, q = s - a * WW;
        const int dy = q / W, dx = q - dy * W;
        const int gr = sr[a] + dy - v, gc = sc[a] + dx - v;
        const int id = (gr >= 0 && gr < h && gc >= 0 && gc < w) ? grid[gr * w + gc] : outside;
        int ncar = 0;
        for (int p = 0; p < N; ++p) ncar += (sr[p] == gr) & (sc[p] == gc);
        tab[s] = make_int2(id, ncar);
    }
    __syncthreads();
    const float inv_vocab = 1.0f / (float)vocab;
    auto value = [&](int a, int off) -> float {     // element `off` of car a's row (TJ:336-362)
        if (!sal[a]) return 0.0f;
        if (off < hdr) return off == 0 ? s0[a] : off == 1 ? s1[a] : off == 2 ? s2[a] : s3[a];
        const int k = off - hdr;
        const int seg = (int)(((float)k + 0.5f) * inv_vocab);
        const int ch = k - seg * vocab;
        const int2 t = tab[a * WW + seg];
        float z = (ch == t.x) ? 1.0f : 0.0f;
        if (ch == car_class) z += (float)t.y;
        return z;
    };
    const int L = N * obs_dim;
    const long long b0 = (long long)e * L;                 // first float of this env in the whole tensor
    const int head = (int)((4 - (b0 & 3)) & 3);
    const int nb = (L - head) >> 2, tail = (L - head) & 3;
    float* out = obs + b0;
    if ((int)threadIdx.x < head) out[threadIdx.x] = value(0, (int)threadIdx.x);                       // head < 4 <= hdr.. row 0
    if ((int)threadIdx.x < tail) {
        const int f = head + 4 * nb + (int)threadIdx.x;
        out[f] = value(f / obs_dim, f % obs_dim);
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4* out4 = reinterpret_cast<f32x4*>(out + head);
    const int o = (int)(((b0 + head) >> 2) & 63);           // 1 KiB-aligned wave stores
    int j = (int)threadIdx.x - o;
    if (j < 0) j += NT;
    int f = head + 4 * j;
    int a = f / obs_dim, off = f - a * obs_dim;
    const int step = 4 * NT, da = step / obs_dim, doff = step - da * obs_dim;
    for (; j < nb; j += NT) {
        f32x4 z;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ai = a, oi = off + i;
            if (oi >= obs_dim) {
                oi -= obs_dim;
                ++ai;
            }
            z[i] = value(ai, oi);
        }
        out4[j] = z;
        a += da;
        off += doff;
        if (off >= obs_dim) {
            off -= obs_dim;
            ++a;
        }
    }
}

// sparse encoder for Traffic-Junction rows (see pp_encode_kernel): enc[a] = bias (dead car: obs row is zero)
// or bias + last_act*Wt[0] + route_frac*Wt[1] + sum_cells ( Wt[2+cell*vocab+id] + ncar*Wt[2+cell*vocab+CAR] ).
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void tj_encode_kernel(TJState st, const f32x4* __restrict__ Wt,
                                                        const f32x4* __restrict__ bias, f32x4* __restrict__ out, int ldo4,
                                                        int H4, const f32x4* __restrict__ loc_table)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x;
    const int N = st.N, W = 2 * st.v + 1, WW = W * W, nseg = N * WW;
    const TJTile t = tj_tile_at(smem, N);
    for (int a = threadIdx.x; a < N; a += blockDim.x) tj_tile_load_car(t, st, e, a);
    __syncthreads();
    for (int q = threadIdx.x; q < nseg; q += blockDim.x) t.tab[q] = tj_tab_entry(t, st, q);
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * H4; idx += blockDim.x) {
        const int a = idx / H4, c4 = idx - a * H4;
        out[((size_t)e * N + a) * ldo4 + c4] = tj_encode_row(t, st, a, c4, H4, Wt, bias, loc_table);
    }
}

__global__ __launch_bounds__(256) void tj_encode_table_kernel(const f32x4* __restrict__ Wt, const int32_t* __restrict__ grid,
                                                              f32x4* __restrict__ table, int h, int w, int v, int vocab,
                                                              int outside, int H4, int hdr)
{
    const int W = 2 * v + 1, WW = W * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w * H4) return;
    const int pos = i / H4, c4 = i - pos * H4;
    f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
    for (int cell = 0; cell < WW; ++cell) {
        const int gr = pos / w + cell / W - v, gc = pos % w + cell % W - v;
        const int id = (gr >= 0 && gr < h && gc >= 0 && gc < w) ? grid[gr * w + gc] : outside;
        if (id >= 0) acc += Wt[((size_t)hdr + (size_t)cell * vocab + id) * H4 + c4];
    }
    table[i] = acc;
}

int tj_encode_table(ic3_env* env, const float* Wt, int H, float* table, hipStream_t s)
{
    const ic3_dims& d = env->dims;
    const int n = d.grid_h * d.grid_w * (H / 4);
    hipLaunchKernelGGL(tj_encode_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const f32x4*>(Wt),
                       env->d_grid, reinterpret_cast<f32x4*>(table), d.grid_h, d.grid_w, env->tj.vision, d.vocab,
                       d.vocab - 3, H / 4, env->tj.vocab_type ? 4 : 2);
    IC3_HIP(hipGetLastError());
    return 0;
}

int tj_group(int N)
{
    const int g = group_lanes(N);
    return g < 8 ? 8 : g;
}

TJState tj_state_of(const ic3_env* env)
{
    const ic3_tj_cfg& c = env->tj;
    const ic3_dims& d = env->dims;
    TJState st;
    st.alive = env->f("alive");
    st.wait = env->f("wait");
    st.loc_r = env->f("loc_r");
    st.loc_c = env->f("loc_c");
    st.last_act = env->f("last_act");
    st.route_loc = env->f("route_loc");
    st.route_id = env->f("route_id");
    st.completed = env->f("is_completed");
    st.cars = env->f("cars_in_sys");
    st.failed = env->f("has_failed");
    st.over = env->f("over");
    st.episode = env->f("episode");
    st.tstep = env->f("t");
    st.route_off = env->d_route_off;
    st.route_rc = env->d_route_rc;
    st.grid = env->d_grid;
    st.thr = env->d_thr;
    st.N = c.N;
    st.narrival = d.narrival;
    st.rpa = d.npath / d.narrival;
    st.h = d.grid_h;
    st.w = d.grid_w;
    st.v = c.vision;
    st.vocab = d.vocab;
    st.outside = d.vocab - 3;
    st.car_class = d.vocab - 1;
    st.npath = d.npath;
    st.hdr = c.vocab_type ? 4 : 2;
    st.seed = c.seed;
    st.gid0 = c.env_id_offset;
    st.ar = AutoReset{ env->auto_max_steps, env->f("episode"), env->f("acc_success"), env->f("acc_episodes"),
                       env->f("acc_steps"), c.seed, c.env_id_offset };
    return st;
}

// The read-only launches (observation rows, sparse encoder): the car fields come from env->view when a snapshot is set
// (ic3_env_observe_at / ic3_env_encode_at), from the live state otherwise.
static TJState tj_view_state_of(const ic3_env* env)
{
    TJState st = tj_state_of(env);
    if (env->view) {
        st.alive = const_cast<int32_t*>(env->fv("alive"));
        st.loc_r = const_cast<int32_t*>(env->fv("loc_r"));
        st.loc_c = const_cast<int32_t*>(env->fv("loc_c"));
        st.last_act = const_cast<int32_t*>(env->fv("last_act"));
        st.route_id = const_cast<int32_t*>(env->fv("route_id"));
        st.route_loc = const_cast<int32_t*>(env->fv("route_loc"));
        st.wait = const_cast<int32_t*>(env->fv("wait"));
    }
    return st;
}

int tj_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table, float* out, int ldo, int H,
              hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    const int WW = env->dims.window * env->dims.window;
    const size_t lds = (size_t)(((7 * c.N + 3) & ~3) + 2 * c.N * WW) * sizeof(int32_t);
    const TJState st = tj_view_state_of(env);
    hipLaunchKernelGGL(tj_encode_kernel, dim3(c.E), dim3(256), lds, s, st, reinterpret_cast<const f32x4*>(Wt),
                       reinterpret_cast<const f32x4*>(bias), reinterpret_cast<f32x4*>(out), ldo / 4, H / 4,
                       reinterpret_cast<const f32x4*>(loc_table));
    IC3_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of tj_encode_kernel (enc_bwd.hpp).  Slots: 0..hdr-1 the header scalars (cols 0..hdr-1), hdr+cell the
// car count of window cell `cell` (col hdr + cell*vocab + car_class).  Dead cars have an all-zero obs row: they
// contribute to dbias only.  Position of a live car = its grid cell.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tj_encode_bwd_kernel(const int32_t* __restrict__ alive_s,
                                                            const int32_t* __restrict__ loc_r,
                                                            const int32_t* __restrict__ loc_c,
                                                            const int32_t* __restrict__ last_act_s,
                                                            const int32_t* __restrict__ route_id_s,
                                                            const float* __restrict__ g, int ldg, float* __restrict__ P,
                                                            float* __restrict__ Dpart, int E, int chunk, int N, int h,
                                                            int w, int v, int npath, int H, int hdr, int tab_words)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int W = 2 * v + 1, WW = W * W, nseg = N * WW, nslots = hdr + WW;
    int32_t* sr = smem;
    int32_t* sc = sr + N;
    int32_t* sal = sc + N;
    float* sh = reinterpret_cast<float*>(sal + N);   // [4][N] header scalars
    int32_t* ncar = smem + ((7 * N + 3) & ~3);        // [nseg]
    float* gl = reinterpret_cast<float*>(smem + tab_words);
    float* Dl = gl + N * H;
    for (int i = threadIdx.x; i < (nslots + 1) * H; i += blockDim.x) Dl[i] = 0.f;
    const int e0 = blockIdx.x * chunk, e1 = min(E, e0 + chunk);
    for (int e = e0; e < e1; ++e) {
        for (int a = threadIdx.x; a < N; a += blockDim.x) {
            const size_t i = (size_t)e * N + a;
            sr[a] = loc_r[i];
            sc[a] = loc_c[i];
            sal[a] = alive_s[i];
            sh[a] = (float)((double)last_act_s[i] / 1.0);                       // same expressions as the forward
            sh[N + a] = (float)((double)route_id_s[i] / (double)(npath - 1));
            sh[2 * N + a] = (float)((double)sr[a] / (double)(h - 1));
            sh[3 * N + a] = (float)((double)sc[a] / (double)(w - 1));
        }
        __syncthreads();
        for (int s = threadIdx.x; s < nseg; s += blockDim.x) {
            const int a = s / WW, q = s - a * WW;
            const int gr = sr[a] + q / W - v, gc = sc[a] + q % W - v;
            int n = 0;
            for (int p = 0; p < N; ++p) n += (sr[p] == gr) & (sc[p] == gc);
            ncar[s] = n;
        }
        __syncthreads();
        enc_bwd_accumulate(
            g, ldg, (size_t)e * N, N, H, gl, Dl, nslots, P,
            [&](int a, int s) {
                if (!sal[a]) return 0.f;
                return s < hdr ? sh[s * N + a] : (float)ncar[a * WW + (s - hdr)];
            },
            [&](int a) { return sal[a] ? sr[a] * w + sc[a] : -1; });
    }
    float* dst = Dpart + (size_t)blockIdx.x * (nslots + 1) * H;
    for (int i = threadIdx.x; i < (nslots + 1) * H; i += blockDim.x) dst[i] = Dl[i];
}

// Stage 1, row-parallel form (enc_bwd.hpp): same slots; a live car's header scalars and the cars inside its window.
__global__ __launch_bounds__(256) void tj_encode_bwd_rows_kernel(const int32_t* __restrict__ alive_s,
                                                                 const int32_t* __restrict__ loc_r,
                                                                 const int32_t* __restrict__ loc_c,
                                                                 const int32_t* __restrict__ last_act_s,
                                                                 const int32_t* __restrict__ route_id_s,
                                                                 const float* __restrict__ g, int ldg,
                                                                 float* __restrict__ Ppart, float* __restrict__ Dpart, int E,
                                                                 int N, int h, int w, int v, int npath, int H, int Hc, int hdr,
                                                                 int accumulate)
{
    IC3_DYNAMIC_LDS(float, smf);
    const int W = 2 * v + 1;
    const int centre = v * W + v;
    enc_bwd_rows(
        g, ldg, E, N, N, H, Hc, h * w, hdr + W * W, Ppart, Dpart, smf,
        [&](size_t i) { return loc_r[i] | (loc_c[i] << 16); },
        [&](const int32_t* ent, int a, size_t row) { return alive_s[row] ? (ent[a] & 0xffff) * w + (ent[a] >> 16) : -1; },
        [&](size_t row, const int32_t* ent, int a, auto reg) {
            if (!alive_s[row]) return -1;                        // an all-zero obs row: bias only
            const int r = ent[a] & 0xffff, c = ent[a] >> 16;
            reg(0, (float)((double)last_act_s[row] / 1.0));                      // same expressions as the forward
            reg(1, (float)((double)route_id_s[row] / (double)(npath - 1)));
            if (hdr == 4) {
                reg(2, (float)((double)r / (double)(h - 1)));
                reg(3, (float)((double)c / (double)(w - 1)));
            }
            reg(4, 1.0f);                                        // the car itself, in its window's centre cell
            return r * w + c;
        },
        [&](const int32_t* ent, int a, int p, size_t row) {      // another car slot standing inside a live car's window
            const int dy = (ent[p] & 0xffff) - (ent[a] & 0xffff) + v, dx = (ent[p] >> 16) - (ent[a] >> 16) + v;
            return (p != a && (unsigned)dy < (unsigned)W && (unsigned)dx < (unsigned)W && alive_s[row]) ? hdr + dy * W + dx : -1;
        },
        [&](int k) { return k < hdr ? k : (k == 4 ? hdr + centre : -1); }, accumulate);
}

// Stage 1, window form (enc_bwd.hpp).
struct TJEncSpec {
    long long off_alive, off_r, off_c, off_act, off_route;   // inside a snapshot, in words
    int total, rows_env, h, w, v, npath, hdr;
    __device__ unsigned word(const int32_t* st, int e, int i) const
    {
        const size_t k = (size_t)e * total + i;
        return (unsigned)st[off_r + k] | ((unsigned)st[off_c + k] << 16);
    }
    static constexpr bool live_always = false;
    static constexpr bool slots_exact_bf16 = false;     // the header scalars are arbitrary fp32
    __device__ bool live(const int32_t* st, size_t row) const { return st[off_alive + row] != 0; }
    __device__ int pos(unsigned wd) const { return (int)(wd & 0xffff) * w + (int)(wd >> 16); }
    template <class Emit>
    __device__ void self(const int32_t* st, size_t row, int, unsigned wd, Emit emit) const
    {
        const int r = (int)(wd & 0xffff), c = (int)(wd >> 16), W = 2 * v + 1;
        emit(0, (float)((double)st[off_act + row] / 1.0));                       // same expressions as the forward
        emit(1, (float)((double)st[off_route + row] / (double)(npath - 1)));
        if (hdr == 4) {
            emit(2, (float)((double)r / (double)(h - 1)));
            emit(3, (float)((double)c / (double)(w - 1)));
        }
        emit(hdr + v * W + v, 1.0f);                             // the car itself, in its window's centre cell
    }
    __device__ int pair(unsigned wa, unsigned wp, int, int) const   // another car slot standing inside a live car's window
    {
        const int W = 2 * v + 1;
        const int dy = (int)(wp & 0xffff) - (int)(wa & 0xffff) + v, dx = (int)(wp >> 16) - (int)(wa >> 16) + v;
        return ((unsigned)dy < (unsigned)W && (unsigned)dx < (unsigned)W) ? hdr + dy * W + dx : -1;
    }
};
template <int MBP>
__global__ __launch_bounds__(256) void tj_encode_bwd_window_kernel(const EncWinArgs a, const TJEncSpec sp)
{
    IC3_DYNAMIC_LDS(unsigned char, sm);
    enc_bwd_window<MBP>(a, sp, sm);
}

__global__ __launch_bounds__(256) void tj_encode_bwd_expand_kernel(const float* __restrict__ P, int np,
                                                                   const float* __restrict__ Dpart, int nwg,
                                                                   const int32_t* __restrict__ grid,
                                                                   float* __restrict__ dWt, float* __restrict__ dbias,
                                                                   int h, int w, int v, int vocab, int outside,
                                                                   int car_class, int H, int hdr)
{
    const int W = 2 * v + 1, WW = W * W, nslots = hdr + WW, npos = h * w;
    const int nsplit = (nwg + ENCB_SPLIT - 1) / ENCB_SPLIT;
    const long long nP = (long long)npos * H, nA = enc_bwd_pfold_threads(nP), nB = (long long)(nslots + 1) * H * nsplit;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < nA) {
            long long ip;
            const float val = enc_bwd_pfold(P, np, (size_t)nP, i, nP, &ip);     // (whole wavefronts take this branch)
            if (ip < 0 || val == 0.f) continue;
            const int c = (int)(ip % H), pos = (int)(ip / H);
            for (int cell = 0; cell < WW; ++cell) {
                const int gr = pos / w + cell / W - v, gc = pos % w + cell % W - v;
                const int id = (gr >= 0 && gr < h && gc >= 0 && gc < w) ? grid[gr * w + gc] : outside;
                if (id >= 0) atomicAdd(dWt + ((size_t)hdr + (size_t)cell * vocab + id) * H + c, val);   // -1: scalar vocab, off-road
            }
        } else {
            const long long j = i - nA;
            const int c = (int)(j % H), s = (int)((j / H) % (nslots + 1)), part = (int)(j / ((long long)H * (nslots + 1)));
            float acc = 0.f;
            const int k1 = min(nwg, (part + 1) * ENCB_SPLIT);
            for (int k = part * ENCB_SPLIT; k < k1; ++k) acc += Dpart[((size_t)k * (nslots + 1) + s) * H + c];
            if (s == nslots) {
                if (dbias) atomicAdd(dbias + c, acc);
            } else {
                const size_t col = s < hdr ? (size_t)s : (size_t)hdr + (size_t)(s - hdr) * vocab + car_class;
                atomicAdd(dWt + col * H + c, acc);
            }
        }
    }
}

int encode_bwd_chunk(int E);
int encode_bwd_items_b(int nwg, int nslots1, int H);

int64_t tj_encode_bwd_work(const ic3_env* env, int H)
{
    const ic3_dims& d = env->dims;
    const int WW = d.window * d.window, hdr = env->tj.vocab_type ? 4 : 2;
    const int chunk = encode_bwd_chunk(env->tj.E), nwg = (env->tj.E + chunk - 1) / chunk;
    const int npos = d.grid_h * d.grid_w;
    const EncBwdPlan pl = enc_bwd_plan(env->tj.E, env->tj.N, env->tj.N, H, npos, hdr + WW);
    const int64_t per_env_form = (int64_t)npos * H + (int64_t)nwg * (hdr + WW + 1) * H;
    const int64_t row_form = pl.csplit ? (int64_t)pl.nrg * (npos + hdr + WW + 1) * H : 0;
    return std::max(per_env_form, row_form);
}

// mode: see pp_encode_bwd
int tj_encode_bwd(ic3_env* env, const int32_t* snap, const float* g, int ldg, int H, float* dWt, float* dbias,
                  float* work, hipStream_t s, int mode)
{
    const ic3_tj_cfg& c = env->tj;
    const ic3_dims& d = env->dims;
    const int WW = d.window * d.window, hdr = c.vocab_type ? 4 : 2;
    const int tab_words = (((7 * c.N + 3) & ~3) + c.N * WW + 3) & ~3;
    const size_t lds = ((size_t)tab_words + (size_t)c.N * H + (size_t)(hdr + WW + 1) * H) * sizeof(int32_t);
    const int32_t* base = snap ? snap : env->state;
    auto fld = [&](const char* name) { return base + (env->f(name) - env->state); };
    const int npos = d.grid_h * d.grid_w;
    const EncBwdPlan pl = enc_bwd_plan(c.E, c.N, c.N, H, npos, hdr + WW);
    if (mode && !pl.csplit) return fail(-38, "ic3_env_encode_backward_accumulate: this configuration takes the per-env form (use ic3_env_encode_backward)");
    if (mode == 0 || mode == 3) {
        IC3_HIP(hipMemsetAsync(dWt, 0, (size_t)d.obs_dim * H * sizeof(float), s));
        if (dbias) IC3_HIP(hipMemsetAsync(dbias, 0, (size_t)H * sizeof(float), s));
    }
    float* P = work;
    int np = 1, nwg;
    float* Dpart;
    if (pl.csplit) {                                   // rows in parallel, P and D of a column slice in LDS
        np = nwg = pl.nrg;
        Dpart = work + (size_t)pl.nrg * npos * H;
        if (mode != 3) {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(tj_encode_bwd_rows_kernel), (size_t)pl.lds));
            hipLaunchKernelGGL(tj_encode_bwd_rows_kernel, dim3(pl.nrg, pl.csplit), dim3(256), pl.lds, s, fld("alive"),
                               fld("loc_r"), fld("loc_c"), fld("last_act"), fld("route_id"), g, ldg, P, Dpart, c.E, c.N, d.grid_h,
                               d.grid_w, c.vision, d.npath, H, H / pl.csplit, hdr, mode == 2 ? 1 : 0);
            IC3_HIP(hipGetLastError());
        }
        if (mode == 1 || mode == 2) return 0;
    } else {
        if (lds > 160 * 1024) return fail(-22, "ic3_env_encode_backward: configuration needs more than 160 KB of LDS");
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(tj_encode_bwd_kernel), lds));
        const int chunk = encode_bwd_chunk(c.E);
        nwg = (c.E + chunk - 1) / chunk;
        Dpart = work + (size_t)npos * H;
        IC3_HIP(hipMemsetAsync(P, 0, (size_t)npos * H * sizeof(float), s));
        hipLaunchKernelGGL(tj_encode_bwd_kernel, dim3(nwg), dim3(256), lds, s, fld("alive"), fld("loc_r"), fld("loc_c"),
                           fld("last_act"), fld("route_id"), g, ldg, P, Dpart, c.E, chunk, c.N, d.grid_h, d.grid_w, c.vision,
                           d.npath, H, hdr, tab_words);
    }
    const long long items = enc_bwd_pfold_threads((long long)npos * H) + encode_bwd_items_b(nwg, hdr + WW + 1, H);
    const int blocks = (int)std::min<long long>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(tj_encode_bwd_expand_kernel, dim3(blocks), dim3(256), 0, s, P, np, Dpart, nwg, env->d_grid, dWt, dbias,
                       d.grid_h, d.grid_w, c.vision, d.vocab, d.vocab - 3, d.vocab - 1, H, hdr);
    IC3_HIP(hipGetLastError());
    return 0;
}

static EncWinPlan tj_win_plan(const ic3_env* env, int H)
{
    const ic3_dims& d = env->dims;
    const int hdr = env->tj.vocab_type ? 4 : 2;
    return enc_win_plan((long long)env->tj.E * env->tj.N, env->tj.N, env->tj.N, H, d.grid_h * d.grid_w, hdr + d.window * d.window, enc_bwd_cus());
}
int64_t tj_encode_bwd_window_work(const ic3_env* env, int H)
{
    const ic3_dims& d = env->dims;
    const int hdr = env->tj.vocab_type ? 4 : 2;
    const EncWinPlan pl = tj_win_plan(env, H);
    return pl.MBP ? (int64_t)pl.nrg * (d.grid_h * d.grid_w + hdr + d.window * d.window + 1) * H : 0;
}
int tj_encode_bwd_window(ic3_env* env, const int32_t* snaps, long long snap_words, int T, const float* g, int ldg, long long g_step,
                         int H, float* work, int first, hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    const ic3_dims& d = env->dims;
    const int hdr = c.vocab_type ? 4 : 2, npos = d.grid_h * d.grid_w, nslots = hdr + d.window * d.window, R = c.E * c.N;
    const EncWinPlan pl = tj_win_plan(env, H);
    if (!pl.MBP || (long long)T * R >= (1ll << 31) - 2 * ENCW_RB) return fail(-38, "ic3_env_encode_backward_window: hid_size a multiple of 32 (and of 128 above 128), T * E * N < 2^31");
    const long long nbat = ((long long)T * R + ENCW_RB - 1) / ENCW_RB;
    EncWinArgs a = { snaps, snap_words, g, g_step, ldg, T, c.E, R, H, npos, nslots, pl.PB, pl.SB, first ? 0 : 1, pl.nstage,
                     (int)((nbat + pl.nrg - 1) / pl.nrg), work, work + (size_t)pl.nrg * npos * H };
    auto off = [&](const char* name) { return (long long)(env->f(name) - env->state); };
    const TJEncSpec sp = { off("alive"), off("loc_r"), off("loc_c"), off("last_act"), off("route_id"), c.N, c.N, d.grid_h, d.grid_w,
                           c.vision, d.npath, hdr };
    const dim3 grid(pl.nrg, pl.ncs, pl.nsl), block(64 * pl.nw);
    auto go = [&](auto kernel) {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (size_t)pl.lds));
        hipLaunchKernelGGL(kernel, grid, block, pl.lds, s, a, sp);
        IC3_HIP(hipGetLastError());
        return 0;
    };
    return pl.MBP == 3 ? go(tj_encode_bwd_window_kernel<3>) : (pl.MBP == 7 ? go(tj_encode_bwd_window_kernel<7>) : go(tj_encode_bwd_window_kernel<13>));
}
int tj_encode_bwd_window_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    const ic3_dims& d = env->dims;
    const int hdr = c.vocab_type ? 4 : 2, npos = d.grid_h * d.grid_w, WW = d.window * d.window;
    const EncWinPlan pl = tj_win_plan(env, H);
    if (!pl.MBP) return fail(-38, "ic3_env_encode_backward_window_finish: this configuration has no window form");
    IC3_HIP(hipMemsetAsync(dWt, 0, (size_t)d.obs_dim * H * sizeof(float), s));
    if (dbias) IC3_HIP(hipMemsetAsync(dbias, 0, (size_t)H * sizeof(float), s));
    const long long items = enc_bwd_pfold_threads((long long)npos * H) + encode_bwd_items_b(pl.nrg, hdr + WW + 1, H);
    const int blocks = (int)std::min<long long>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(tj_encode_bwd_expand_kernel, dim3(blocks), dim3(256), 0, s, work, pl.nrg, work + (size_t)pl.nrg * npos * H, pl.nrg,
                       env->d_grid, dWt, dbias, d.grid_h, d.grid_w, c.vision, d.vocab, d.vocab - 3, d.vocab - 1, H, hdr);
    IC3_HIP(hipGetLastError());
    return 0;
}

__global__ void set_i32_kernel(int32_t* p, int32_t v) { *p = v; }

int tj_reset(ic3_env* env, hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, s, env->d_thr, (int32_t)tj_rate_threshold(env->add_rate));
    const int n = c.E * c.N;
    hipLaunchKernelGGL(tj_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, env->f("alive"), env->f("wait"),
                       env->f("loc_r"), env->f("loc_c"), env->f("last_act"), env->f("route_loc"), env->f("route_id"),
                       env->f("is_completed"), env->f("cars_in_sys"), env->f("has_failed"), env->f("over"),
                       env->f("episode"), env->f("t"), c.E, c.N);
    IC3_HIP(hipGetLastError());
    return 0;
}

int tj_step(ic3_env* env, const int32_t* actions, float* reward, int32_t* done, int32_t* alive, int32_t* is_completed,
            hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    const int G = tj_group(c.N);
    const long long threads = (long long)c.E * G;
    const StepOut out = { reward, done, alive, is_completed, env->d_err };
    hipLaunchKernelGGL(tj_step_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, tj_state_of(env), out, actions,
                       c.E, G);
    IC3_HIP(hipGetLastError());
    return 0;
}

// Small rows (chunks below 8192 floats, e.g. TJ-medium: 5 330 floats = 21 KB per env): the rows are ~96 % zeros, so the
// env's chunk is zero-filled with 1 KiB-aligned float4 wave stores (<= 3 ragged floats at either end as dwords) and the
// <= 2 + 2 W^2 non-zero entries per live car are patched in behind a barrier — 1/4 of the store instructions of the
// element-wise dword kernel and no per-element evaluation.  Same bytes, same values (tests: bit-equal to the oracle).
__global__ __launch_bounds__(1024) void tj_obs_fill_kernel(TJState st, float* __restrict__ obs, int obs_dim, int WW)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x, NT = blockDim.x, tid = threadIdx.x, N = st.N;
    const TJTile t = tj_tile_at(smem, N);
    const int L = N * obs_dim;
    const long long b0 = (long long)e * L;                 // first float of this env in the whole tensor
    float* out = obs + b0;
    // zero fill first: it does not depend on the state, so its stores run under the descriptor loads below
    const int head = (int)((4 - (b0 & 3)) & 3);
    const int nb = (L - head) >> 2, tail = (L - head) & 3;
    if (tid < head) out[tid] = 0.0f;
    if (tid < tail) out[head + 4 * nb + tid] = 0.0f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4* out4 = reinterpret_cast<f32x4*>(out + head);
    const int o = (int)(((b0 + head) >> 2) & 63);           // 1 KiB-aligned wave stores
    int j = tid - o;
    if (j < 0) j += NT;
    const f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
    for (; j < nb; j += NT) out4[j] = z4;
    for (int a = tid; a < N; a += NT) tj_tile_load_car(t, st, e, a);
    __syncthreads();
    for (int q = tid; q < N * WW; q += NT) t.tab[q] = tj_tab_entry(t, st, q);
    __syncthreads();                                        // (also: every zero store of the workgroup has completed)
    for (int q = tid; q < N + N * WW; q += NT) tj_obs_patch(t, st, out, obs_dim, WW, q);
}

int tj_observe(ic3_env* env, float* obs, hipStream_t s)
{
    const ic3_tj_cfg& c = env->tj;
    const ic3_dims& d = env->dims;
    const int WW = d.window * d.window;
    const size_t lds = (size_t)(((7 * c.N + 3) & ~3) + 2 * c.N * WW) * sizeof(int32_t);
    static const int fill = getenv("IC3_TJ_OBS_FILL") ? atoi(getenv("IC3_TJ_OBS_FILL")) : 1;
    const bool big = (long long)c.N * d.obs_dim >= 8192 && d.obs_dim >= 8;
    if (fill == 2 && big) {   // experiment: zero fill + patch for large chunks too, 1024 threads per env
        hipLaunchKernelGGL(tj_obs_fill_kernel, dim3(c.E), dim3(1024), lds, s, tj_view_state_of(env), obs, d.obs_dim, WW);
        IC3_HIP(hipGetLastError());
        return 0;
    }
    // measured on MI355X in the rollout loop: TJ-hard v1 (26 500 floats/env) 5.64 TB/s with the float4 kernel vs 5.12
    // with dword stores; TJ-medium v1 (5 330 floats/env) 3.83 vs 4.32 -> float4 only for large chunks
    if (big) {
        hipLaunchKernelGGL(tj_obs_vec4_kernel, dim3(c.E), dim3(1024), lds, s, env->fv("alive"), env->fv("loc_r"),
                           env->fv("loc_c"), env->fv("last_act"), env->fv("route_id"), env->d_grid, obs, c.N, d.grid_h,
                           d.grid_w, c.vision, d.vocab, d.vocab - 3, d.vocab - 1, d.npath, c.vocab_type ? 4 : 2);
        IC3_HIP(hipGetLastError());
        return 0;
    }
    // small rows: zero fill + patches, 256 threads per env (TJ-medium: 5.66 TB/s = 0.71 of peak; the element-wise dword
    // kernel below, kept as IC3_TJ_OBS_FILL=0, reached 4.35)
    if (fill) {
        hipLaunchKernelGGL(tj_obs_fill_kernel, dim3(c.E), dim3(256), lds, s, tj_view_state_of(env), obs, d.obs_dim, WW);
        IC3_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(tj_obs_kernel, dim3(c.E), dim3(256), lds, s, env->fv("alive"), env->fv("loc_r"), env->fv("loc_c"),
                       env->fv("last_act"), env->fv("route_id"), env->d_grid, obs, c.N, d.grid_h, d.grid_w, c.vision,
                       d.vocab, d.vocab - 3, d.vocab - 1, d.npath, c.vocab_type ? 4 : 2);
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // namespace ic3
