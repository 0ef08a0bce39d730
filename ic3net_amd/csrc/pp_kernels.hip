// pp_kernels.hip — Predator-Prey: reset / step / observation assembly for E environments (gfx950).
//
// Reference semantics: /root/reference/ic3net-envs/ic3net_envs/predator_prey_env.py (cited "PP:line").
// State is struct-of-arrays in HBM, one int32 array per field, env-major ([e][n]) so that the lane
// mapping (env, agent) -> consecutive lanes reads/writes consecutive words.
#include <algorithm>

#include "enc_bwd.hpp"
#include "env_device.hpp"
#include "ic3_common.hpp"

namespace ic3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: one global_store_dwordx4

// ------------------------------------------------------------------------------------------------
// reset: PP:146-168 + _get_cordinates PP:173-175 (np.random.choice(dim*dim, N+nprey, replace=False))
// = sequential rejection sampling of distinct cells on the injected stream.  One lane per env (the
// draw sequence is inherently serial; runs once per episode).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_reset_kernel(int32_t* __restrict__ loc_r, int32_t* __restrict__ loc_c,
                                                       int32_t* __restrict__ reached, int32_t* __restrict__ over,
                                                       int32_t* __restrict__ success, int32_t* __restrict__ episode,
                                                       int32_t* __restrict__ tstep, int E, int N, int nprey, int dim,
                                                       uint32_t seed, uint32_t gid0)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int total = N + nprey;
    const uint32_t ep = (uint32_t)(episode[e] + 1);
    pp_place_entities(loc_r + (size_t)e * total, loc_c + (size_t)e * total, total, dim, seed, gid0 + (uint32_t)e, ep);
    for (int i = 0; i < N; ++i) reached[(size_t)e * N + i] = 0;  // PP:155
    over[e] = 0;                                                  // PP:154
    success[e] = 0;
    episode[e] = (int32_t)ep;
    tstep[e] = 0;
}

// ------------------------------------------------------------------------------------------------
// step: PP:112-144 = _take_action for every predator (PP:212-252), then _get_reward (PP:254-290).
// G = pow2 >= N lanes per env; the per-env reductions (predators on prey, all reached) are wave
// ballots restricted to the env's lane group.
// ------------------------------------------------------------------------------------------------
// The step body lives in env_device.hpp (pp_step_lanes): the fused policy+step kernel runs the same code.
__global__ __launch_bounds__(256) void pp_step_kernel(PPState st, StepOut out, const int32_t* __restrict__ actions, int E,
                                                      int G)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / G, n = tid - e * G;
    pp_step_lanes(st, out, e, n, E, G, [&]() { return actions[(size_t)e * st.rows + n]; });
}

// ------------------------------------------------------------------------------------------------
// observation assembly: PP:188-210 (+ one-hot base grid PP:177-186, flatten env_wrappers.py:88-100).
// The reference copies a (dim+2v)^2 x vocab int64 one-hot grid per step and slices windows; here each
// workgroup owns one env (its N rows are contiguous: N*obs_dim floats), stages the N+nprey positions
// and the per-window-cell descriptors in LDS, and streams the rows out exactly once with 16-byte
// stores.  This is the HBM-write-bound kernel the roofline is quoted on: algorithmic bytes per env =
// N * obs_dim * 4.
//   row a, window cell s = dy*W+dx, channel ch:  obs[a][s*vocab + ch] =
//       (ch == id(s))            id = r*dim+c inside the grid, OUTSIDE_CLASS = dim*dim+1 otherwise
//     + (ch == PREY_CLASS)  * #prey on the cell       (counts, quirk Q3)
//     + (ch == PREDATOR_CLASS) * #predators on the cell
// ------------------------------------------------------------------------------------------------
// Per-env window descriptors in LDS: tab[a*W*W + dy*W + dx] = (one-hot channel, #predators | #prey << 16)
// for agent a's window cell (dy, dx).  Shared by the obs-assembly and the sparse-encoder kernels.
__device__ __forceinline__ const int2* pp_build_tab(int32_t* smem, const int32_t* __restrict__ loc_r,
                                                    const int32_t* __restrict__ loc_c, int e, int N, int nprey,
                                                    int dim, int v, int rows)
{
    // rows = observed agents: the N predators, plus the prey with enemy_comm (PP:203-207); entity a's window is
    // centred on loc[a] either way (prey positions follow the predators in the loc arrays)
    const int total = N + nprey, W = 2 * v + 1, nseg = rows * W * W;
    int32_t* sr = smem;              // [total]
    int32_t* sc = sr + total;        // [total]
    int2* tab = reinterpret_cast<int2*>(smem + ((2 * total + 3) & ~3));  // [nseg]
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        sr[i] = loc_r[(size_t)e * total + i];
        sc[i] = loc_c[(size_t)e * total + i];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nseg; s += blockDim.x) tab[s] = pp_tab_entry(sr, sc, s, N, total, dim, v);
    __syncthreads();
    return tab;
}

template <bool VEC4, bool NT, bool ALIGN, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void pp_obs_kernel(const int32_t* __restrict__ loc_r,
                                                     const int32_t* __restrict__ loc_c, float* __restrict__ obs,
                                                     int N, int nprey, int dim, int v, int rows)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x;
    const int W = 2 * v + 1, nseg = rows * W * W;
    const int vocab = dim * dim + 4;
    const int2* tab = pp_build_tab(smem, loc_r, loc_c, e, N, nprey, dim, v, rows);

    if constexpr (VEC4) {
        // vocab % 4 == 0: a 16-byte store never straddles a window cell; the three special channels
        // (OUTSIDE, PREY, PREDATOR = vocab-3, -2, -1) share the cell's last float4.
        const int segq = vocab >> 2;
        const int Q = nseg * segq;                         // float4s of this env (contiguous rows)
        f32x4* out = reinterpret_cast<f32x4*>(obs + (size_t)e * nseg * vocab);
        // An env chunk is Q*16 bytes, in general not a multiple of the 1 KiB a wavefront stores per
        // instruction: shift the lane->float4 mapping by o = (e*Q) mod 64 so that every wave-store is
        // 1 KiB-aligned in the global address space (no partial cache lines except at chunk ends).
        const int o = ALIGN ? (int)(((long long)e * Q) & 63) : 0;
        int g = (int)threadIdx.x - o;
        if (g < 0) g += THREADS;
        int seg = g / segq, q = g - seg * segq;
        const int dseg = THREADS / segq, dq = THREADS - dseg * segq;
        for (; g < Q; g += THREADS) {
            const int2 t = tab[seg];
            f32x4 z = { 0.f, 0.f, 0.f, 0.f };
            if ((t.x >> 2) == q) {
                const int j = t.x & 3;
                z.x = (j == 0) ? 1.f : 0.f;
                z.y = (j == 1) ? 1.f : 0.f;
                z.z = (j == 2) ? 1.f : 0.f;
                z.w = (j == 3) ? 1.f : 0.f;
            }
            if (q == segq - 1) {
                z.z += (float)(t.y >> 16);
                z.w += (float)(t.y & 0xffff);
            }
            if constexpr (NT) __builtin_nontemporal_store(z, out + g);
            else out[g] = z;
            seg += dseg;
            q += dq;
            if (q >= segq) {
                q -= segq;
                ++seg;
            }
        }
    } else {
        const int total_f = nseg * vocab;
        float* out = obs + (size_t)e * total_f;
        int seg = threadIdx.x / vocab, ch = threadIdx.x - seg * vocab;
        const int dseg = THREADS / vocab, dch = THREADS - dseg * vocab;
        for (int g = threadIdx.x; g < total_f; g += THREADS) {
            const int2 t = tab[seg];
            float z = (ch == t.x) ? 1.f : 0.f;
            if (ch == vocab - 2) z += (float)(t.y >> 16);
            if (ch == vocab - 1) z += (float)(t.y & 0xffff);
            out[g] = z;
            seg += dseg;
            ch += dch;
            if (ch >= vocab) {
                ch -= vocab;
                ++seg;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sparse encoder: nn.Linear(obs_dim, H) of comm.py:51,119 applied to the observation WITHOUT reading
// it back: a row of obs has <= 3 non-zeros per window cell, so  enc[a] = bias + sum_cells ( Wt[cell*vocab+id]
// + npred * Wt[cell*vocab+PRED] + nprey * Wt[cell*vocab+PREY] )  with Wt = encoder.weight^T ([obs_dim][H]).
// One workgroup per env; each group of H/4 lanes owns one agent and reads whole 4H-byte rows of Wt (L2
// resident: obs_dim*H*4 = 1.9 MB for PP-hard).  Replaces a 2*N*obs_dim*H-flop dense fp32 GEMM per env.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_encode_kernel(const int32_t* __restrict__ loc_r,
                                                        const int32_t* __restrict__ loc_c,
                                                        const f32x4* __restrict__ Wt, const f32x4* __restrict__ bias,
                                                        f32x4* __restrict__ out, int ldo4, int N, int nprey, int dim,
                                                        int v, int H4, int rows, const f32x4* __restrict__ loc_table)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int e = blockIdx.x;
    const int WW = (2 * v + 1) * (2 * v + 1);
    const int vocab = dim * dim + 4;
    const int2* tab = pp_build_tab(smem, loc_r, loc_c, e, N, nprey, dim, v, rows);
    for (int idx = threadIdx.x; idx < rows * H4; idx += blockDim.x) {
        const int a = idx / H4, c4 = idx - a * H4;
        out[((size_t)e * rows + a) * ldo4 + c4] =
            pp_encode_row(smem, smem + (N + nprey), tab, a, c4, H4, WW, vocab, dim, Wt, bias, loc_table);
    }
}

// loc_table[pos] = sum_cells Wt[cell*vocab + id(pos, cell)]  (same cell order as the gather it replaces)
__global__ __launch_bounds__(256) void pp_encode_table_kernel(const f32x4* __restrict__ Wt, f32x4* __restrict__ table,
                                                              int dim, int v, int H4)
{
    const int W = 2 * v + 1, WW = W * W, vocab = dim * dim + 4, OUTSIDE = dim * dim + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim * dim * H4) return;
    const int pos = i / H4, c4 = i - pos * H4;
    f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
    for (int cell = 0; cell < WW; ++cell) {
        const int gr = pos / dim + cell / W - v, gc = pos % dim + cell % W - v;
        const int id = (gr >= 0 && gr < dim && gc >= 0 && gc < dim) ? gr * dim + gc : OUTSIDE;
        acc += Wt[((size_t)cell * vocab + id) * H4 + c4];
    }
    table[i] = acc;
}

int pp_encode_table(ic3_env* env, const float* Wt, int H, float* table, hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int n = c.dim * c.dim * (H / 4);
    hipLaunchKernelGGL(pp_encode_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const f32x4*>(Wt),
                       reinterpret_cast<f32x4*>(table), c.dim, c.vision, H / 4);
    IC3_HIP(hipGetLastError());
    return 0;
}

int pp_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table, float* out, int ldo, int H,
              hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int rows = env->dims.N;
    const int total = c.N + c.nprey, W = 2 * c.vision + 1, nseg = rows * W * W;
    const size_t lds = (size_t)(((2 * total + 3) & ~3) + 2 * nseg) * sizeof(int32_t);
    hipLaunchKernelGGL(pp_encode_kernel, dim3(c.E), dim3(256), lds, s, env->fv("loc_r"), env->fv("loc_c"),   // (view-aware)
                       reinterpret_cast<const f32x4*>(Wt), reinterpret_cast<const f32x4*>(bias),
                       reinterpret_cast<f32x4*>(out), ldo / 4, c.N, c.nprey, c.dim, c.vision, H / 4, rows,
                       reinterpret_cast<const f32x4*>(loc_table));
    IC3_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of pp_encode_kernel (enc_bwd.hpp).  Slots: 2*cell -> PREDATOR count (col cell*vocab + vocab-1),
// 2*cell+1 -> PREY count (col cell*vocab + vocab-2); position of row a = loc[a] (always on the grid).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pp_encode_bwd_kernel(const int32_t* __restrict__ loc_r,
                                                            const int32_t* __restrict__ loc_c,
                                                            const float* __restrict__ g, int ldg, float* __restrict__ P,
                                                            float* __restrict__ Dpart, int E, int chunk, int N, int nprey,
                                                            int dim, int v, int H, int rows, int tab_words)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    const int WW = (2 * v + 1) * (2 * v + 1), nslots = 2 * WW;
    float* gl = reinterpret_cast<float*>(smem + tab_words);
    float* Dl = gl + rows * H;
    for (int i = threadIdx.x; i < (nslots + 1) * H; i += blockDim.x) Dl[i] = 0.f;
    const int e0 = blockIdx.x * chunk, e1 = min(E, e0 + chunk);
    for (int e = e0; e < e1; ++e) {
        const int2* tab = pp_build_tab(smem, loc_r, loc_c, e, N, nprey, dim, v, rows);   // ends with a barrier
        const int32_t* sr = smem;
        const int32_t* sc = smem + (N + nprey);
        enc_bwd_accumulate(
            g, ldg, (size_t)e * rows, rows, H, gl, Dl, nslots, P,
            [&](int a, int s) {
                const int y = tab[a * WW + (s >> 1)].y;
                return (float)((s & 1) ? (y >> 16) : (y & 0xffff));
            },
            [&](int a) { return sr[a] * dim + sc[a]; });
    }
    float* dst = Dpart + (size_t)blockIdx.x * (nslots + 1) * H;
    for (int i = threadIdx.x; i < (nslots + 1) * H; i += blockDim.x) dst[i] = Dl[i];
}

// Stage 1, row-parallel form (enc_bwd.hpp): slot 2*cell + 0 / 1 = predators / prey seen in window cell `cell`.
__global__ __launch_bounds__(256) void pp_encode_bwd_rows_kernel(const int32_t* __restrict__ loc_r,
                                                                 const int32_t* __restrict__ loc_c,
                                                                 const float* __restrict__ g, int ldg,
                                                                 float* __restrict__ Ppart, float* __restrict__ Dpart, int E,
                                                                 int N, int nprey, int dim, int v, int H, int Hc, int rows,
                                                                 int accumulate)
{
    IC3_DYNAMIC_LDS(float, smf);
    const int W = 2 * v + 1, total = N + nprey;
    const int centre = v * W + v;
    enc_bwd_rows(
        g, ldg, E, rows, total, H, Hc, dim * dim, 2 * W * W, Ppart, Dpart, smf,
        [&](size_t i) { return loc_r[i] | (loc_c[i] << 16); },
        [&](const int32_t* ent, int a, size_t) { return (ent[a] & 0xffff) * dim + (ent[a] >> 16); },
        [&](size_t, const int32_t* ent, int a, auto reg) {
            reg(0, a < N ? 1.0f : 0.0f);                         // the observer itself, in its window's centre cell
            reg(1, a < N ? 0.0f : 1.0f);
            return (ent[a] & 0xffff) * dim + (ent[a] >> 16);
        },
        [&](const int32_t* ent, int a, int p, size_t) {          // PP:191-195: another entity standing inside the window
            const int dy = (ent[p] & 0xffff) - (ent[a] & 0xffff) + v, dx = (ent[p] >> 16) - (ent[a] >> 16) + v;
            return (p != a && (unsigned)dy < (unsigned)W && (unsigned)dx < (unsigned)W) ? 2 * (dy * W + dx) + (p >= N ? 1 : 0)
                                                                                         : -1;
        },
        [&](int k) { return k == 0 ? 2 * centre : (k == 1 ? 2 * centre + 1 : -1); }, accumulate);
}

// Stage 1, window form (enc_bwd.hpp: a window of recorded states in one launch, position sums as one-hot products).
struct PPEncSpec {
    long long off_r, off_c;              // loc_r / loc_c inside a snapshot, in words
    int N, total, rows_env, dim, v;
    __device__ unsigned word(const int32_t* st, int e, int i) const
    {
        const size_t k = (size_t)e * total + i;
        return (unsigned)st[off_r + k] | ((unsigned)st[off_c + k] << 16);
    }
    static constexpr bool live_always = true;
    static constexpr bool slots_exact_bf16 = true;     // entity counts
    __device__ bool live(const int32_t*, size_t) const { return true; }
    __device__ int pos(unsigned w) const { return (int)(w & 0xffff) * dim + (int)(w >> 16); }
    template <class Emit>
    __device__ void self(const int32_t*, size_t, int a, unsigned, Emit emit) const
    {
        const int W = 2 * v + 1;
        emit(2 * (v * W + v) + (a < N ? 0 : 1), 1.0f);           // the observer itself, in its window's centre cell
    }
    __device__ int pair(unsigned wa, unsigned wp, int, int p) const   // PP:191-195: another entity standing inside the window
    {
        const int W = 2 * v + 1;
        const int dy = (int)(wp & 0xffff) - (int)(wa & 0xffff) + v, dx = (int)(wp >> 16) - (int)(wa >> 16) + v;
        return ((unsigned)dy < (unsigned)W && (unsigned)dx < (unsigned)W) ? 2 * (dy * W + dx) + (p >= N ? 1 : 0) : -1;
    }
};
template <int MBP>
__global__ __launch_bounds__(256) void pp_encode_bwd_window_kernel(const EncWinArgs a, const PPEncSpec sp)
{
    IC3_DYNAMIC_LDS(unsigned char, sm);
    enc_bwd_window<MBP>(a, sp, sm);
}

// Stage 2 for PP: dWt (zeroed by the caller) += P through the id map, class columns and dbias from the partials.
// P = the sum of `np` partials of npos * H floats each.
__global__ __launch_bounds__(256) void pp_encode_bwd_expand_kernel(const float* __restrict__ P, int np,
                                                                   const float* __restrict__ Dpart, int nwg,
                                                                   float* __restrict__ dWt, float* __restrict__ dbias,
                                                                   int dim, int v, int H)
{
    const int W = 2 * v + 1, WW = W * W, nslots = 2 * WW, npos = dim * dim, vocab = dim * dim + 4;
    const int OUTSIDE = dim * dim + 1;
    // partials: ENCB_SPLIT of them per thread, so that the reduction over workgroups is spread over many threads
    const int nsplit = (nwg + ENCB_SPLIT - 1) / ENCB_SPLIT;
    const long long nP = (long long)npos * H, nA = enc_bwd_pfold_threads(nP), nB = (long long)(nslots + 1) * H * nsplit;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < nA) {
            long long ip;
            const float val = enc_bwd_pfold(P, np, (size_t)nP, i, nP, &ip);     // (whole wavefronts take this branch)
            if (ip < 0 || val == 0.f) continue;
            const int h = (int)(ip % H), pos = (int)(ip / H);
            for (int cell = 0; cell < WW; ++cell) {
                const int gr = pos / dim + cell / W - v, gc = pos % dim + cell % W - v;
                const int id = (gr >= 0 && gr < dim && gc >= 0 && gc < dim) ? gr * dim + gc : OUTSIDE;
                atomicAdd(dWt + ((size_t)cell * vocab + id) * H + h, val);
            }
        } else {
            const long long j = i - nA;
            const int h = (int)(j % H), s = (int)((j / H) % (nslots + 1)), part = (int)(j / ((long long)H * (nslots + 1)));
            float acc = 0.f;
            const int w1 = min(nwg, (part + 1) * ENCB_SPLIT);
            for (int w = part * ENCB_SPLIT; w < w1; ++w) acc += Dpart[((size_t)w * (nslots + 1) + s) * H + h];
            if (s == nslots) {
                if (dbias) atomicAdd(dbias + h, acc);
            } else {
                const int col = (s >> 1) * vocab + ((s & 1) ? vocab - 2 : vocab - 1);
                atomicAdd(dWt + (size_t)col * H + h, acc);
            }
        }
    }
}

// envs per workgroup of the accumulate stage: 2048 workgroups (8 per CU at ~15 KB of LDS each) instead of 512 — a workgroup
// walks its envs one after the other with three barriers each, so what hides that latency is more workgroups per CU
int encode_bwd_chunk(int E) { return (E + 2047) / 2048; }
int encode_bwd_items_b(int nwg, int nslots1, int H) { return nslots1 * H * ((nwg + ENCB_SPLIT - 1) / ENCB_SPLIT); }

int64_t pp_encode_bwd_work(const ic3_env* env, int H)
{
    const ic3_pp_cfg& c = env->pp;
    const int WW = (2 * c.vision + 1) * (2 * c.vision + 1);
    const int chunk = encode_bwd_chunk(c.E), nwg = (c.E + chunk - 1) / chunk;
    const EncBwdPlan pl = enc_bwd_plan(c.E, env->dims.N, c.N + c.nprey, H, c.dim * c.dim, 2 * WW);
    const int64_t per_env_form = (int64_t)c.dim * c.dim * H + (int64_t)nwg * (2 * WW + 1) * H;
    const int64_t row_form = pl.csplit ? (int64_t)pl.nrg * (c.dim * c.dim + 2 * WW + 1) * H : 0;
    return std::max(per_env_form, row_form);
}

// mode 0: both stages (ic3_env_encode_backward); 1 / 2: stage 1 writing / adding to the partials in `work`
// (ic3_env_encode_backward_accumulate); 3: stage 2 alone (ic3_env_encode_backward_finish)
int pp_encode_bwd(ic3_env* env, const int32_t* snap, const float* g, int ldg, int H, float* dWt, float* dbias,
                  float* work, hipStream_t s, int mode)
{
    const ic3_pp_cfg& c = env->pp;
    const int rows = env->dims.N;
    const int total = c.N + c.nprey, W = 2 * c.vision + 1, WW = W * W, nseg = rows * WW;
    const int tab_words = (((2 * total + 3) & ~3) + 2 * nseg + 3) & ~3;
    const size_t lds = ((size_t)tab_words + (size_t)rows * H + (size_t)(2 * WW + 1) * H) * sizeof(int32_t);
    const int32_t* base = snap ? snap : env->state;
    const int32_t* loc_r = base + (env->f("loc_r") - env->state);
    const int32_t* loc_c = base + (env->f("loc_c") - env->state);
    const int npos = c.dim * c.dim;
    const EncBwdPlan pl = enc_bwd_plan(c.E, rows, total, H, npos, 2 * WW);
    if (mode && !pl.csplit) return fail(-38, "ic3_env_encode_backward_accumulate: this configuration takes the per-env form (use ic3_env_encode_backward)");
    if (mode == 0 || mode == 3) {
        IC3_HIP(hipMemsetAsync(dWt, 0, (size_t)env->dims.obs_dim * H * sizeof(float), s));
        if (dbias) IC3_HIP(hipMemsetAsync(dbias, 0, (size_t)H * sizeof(float), s));
    }
    float* P = work;
    int np = 1, nwg;
    float* Dpart;
    if (pl.csplit) {                                   // rows in parallel, P and D of a column slice in LDS
        np = nwg = pl.nrg;
        Dpart = work + (size_t)pl.nrg * npos * H;
        if (mode != 3) {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(pp_encode_bwd_rows_kernel), (size_t)pl.lds));
            hipLaunchKernelGGL(pp_encode_bwd_rows_kernel, dim3(pl.nrg, pl.csplit), dim3(256), pl.lds, s, loc_r, loc_c, g, ldg, P,
                               Dpart, c.E, c.N, c.nprey, c.dim, c.vision, H, H / pl.csplit, rows, mode == 2 ? 1 : 0);
            IC3_HIP(hipGetLastError());
        }
        if (mode == 1 || mode == 2) return 0;
    } else {                                           // the grid does not fit in LDS: one env at a time, P by global atomics
        if (lds > 160 * 1024) return fail(-22, "ic3_env_encode_backward: configuration needs more than 160 KB of LDS");
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(pp_encode_bwd_kernel), lds));
        const int chunk = encode_bwd_chunk(c.E);
        nwg = (c.E + chunk - 1) / chunk;
        Dpart = work + (size_t)npos * H;
        IC3_HIP(hipMemsetAsync(P, 0, (size_t)npos * H * sizeof(float), s));
        hipLaunchKernelGGL(pp_encode_bwd_kernel, dim3(nwg), dim3(256), lds, s, loc_r, loc_c, g, ldg, P, Dpart, c.E, chunk,
                           c.N, c.nprey, c.dim, c.vision, H, rows, tab_words);
    }
    const long long items = enc_bwd_pfold_threads((long long)npos * H) + encode_bwd_items_b(nwg, 2 * WW + 1, H);
    const int blocks = (int)std::min<long long>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(pp_encode_bwd_expand_kernel, dim3(blocks), dim3(256), 0, s, P, np, Dpart, nwg, dWt, dbias, c.dim,
                       c.vision, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

int enc_bwd_cus()
{
    static int cus[64] = {};      // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}
static EncWinPlan pp_win_plan(const ic3_env* env, int H)
{
    const ic3_pp_cfg& c = env->pp;
    const int W = 2 * c.vision + 1;
    return enc_win_plan((long long)c.E * env->dims.N, env->dims.N, c.N + c.nprey, H, c.dim * c.dim, 2 * W * W, enc_bwd_cus());
}
int64_t pp_encode_bwd_window_work(const ic3_env* env, int H)
{
    const ic3_pp_cfg& c = env->pp;
    const int W = 2 * c.vision + 1;
    const EncWinPlan pl = pp_win_plan(env, H);
    return pl.MBP ? (int64_t)pl.nrg * (c.dim * c.dim + 2 * W * W + 1) * H : 0;
}
// g of (t, row) at g + t * g_step + row * ldg, the state of step t at snaps + t * snap_words; first != 0 writes the partials in
// `work`, 0 adds to them.  finish: the expand stage over those partials.
int pp_encode_bwd_window(ic3_env* env, const int32_t* snaps, long long snap_words, int T, const float* g, int ldg, long long g_step,
                         int H, float* work, int first, hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int W = 2 * c.vision + 1, npos = c.dim * c.dim, nslots = 2 * W * W, R = c.E * env->dims.N;
    const EncWinPlan pl = pp_win_plan(env, H);
    if (!pl.MBP || (long long)T * R >= (1ll << 31) - 2 * ENCW_RB) return fail(-38, "ic3_env_encode_backward_window: hid_size a multiple of 32 (and of 128 above 128), T * E * N < 2^31");
    const long long nbat = ((long long)T * R + ENCW_RB - 1) / ENCW_RB;
    EncWinArgs a = { snaps, snap_words, g, g_step, ldg, T, c.E, R, H, npos, nslots, pl.PB, pl.SB, first ? 0 : 1, pl.nstage,
                     (int)((nbat + pl.nrg - 1) / pl.nrg), work, work + (size_t)pl.nrg * npos * H };
    const PPEncSpec sp = { env->f("loc_r") - env->state, env->f("loc_c") - env->state, c.N, c.N + c.nprey, env->dims.N, c.dim, c.vision };
    const dim3 grid(pl.nrg, pl.ncs, pl.nsl), block(64 * pl.nw);
    auto go = [&](auto kernel) {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (size_t)pl.lds));
        hipLaunchKernelGGL(kernel, grid, block, pl.lds, s, a, sp);
        IC3_HIP(hipGetLastError());
        return 0;
    };
    return pl.MBP == 3 ? go(pp_encode_bwd_window_kernel<3>) : (pl.MBP == 7 ? go(pp_encode_bwd_window_kernel<7>) : go(pp_encode_bwd_window_kernel<13>));
}
int pp_encode_bwd_window_finish(ic3_env* env, int H, float* dWt, float* dbias, float* work, hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int W = 2 * c.vision + 1, npos = c.dim * c.dim;
    const EncWinPlan pl = pp_win_plan(env, H);
    if (!pl.MBP) return fail(-38, "ic3_env_encode_backward_window_finish: this configuration has no window form");
    IC3_HIP(hipMemsetAsync(dWt, 0, (size_t)env->dims.obs_dim * H * sizeof(float), s));
    if (dbias) IC3_HIP(hipMemsetAsync(dbias, 0, (size_t)H * sizeof(float), s));
    const long long items = enc_bwd_pfold_threads((long long)npos * H) + encode_bwd_items_b(pl.nrg, 2 * W * W + 1, H);
    const int blocks = (int)std::min<long long>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(pp_encode_bwd_expand_kernel, dim3(blocks), dim3(256), 0, s, work, pl.nrg, work + (size_t)pl.nrg * npos * H, pl.nrg,
                       dWt, dbias, c.dim, c.vision, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

int pp_reset(ic3_env* env, hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int blocks = (c.E + 255) / 256;
    hipLaunchKernelGGL(pp_reset_kernel, dim3(blocks), dim3(256), 0, s, env->f("loc_r"), env->f("loc_c"),
                       env->f("reached"), env->f("over"), env->f("success"), env->f("episode"), env->f("t"), c.E, c.N,
                       c.nprey, c.dim, c.seed, c.env_id_offset);
    IC3_HIP(hipGetLastError());
    return 0;
}

PPState pp_state_of(const ic3_env* env)
{
    const ic3_pp_cfg& c = env->pp;
    PPState st;
    st.loc_r = env->f("loc_r");
    st.loc_c = env->f("loc_c");
    st.reached = env->f("reached");
    st.over = env->f("over");
    st.success = env->f("success");
    st.tstep = env->f("t");
    st.Np = c.N;
    st.nprey = c.nprey;
    st.dim = c.dim;
    st.v = c.vision;
    st.mode = c.mode;
    st.naction = c.stay ? 5 : 4;
    st.rows = env->dims.N;
    st.ar = AutoReset{ env->auto_max_steps, env->f("episode"), env->f("acc_success"), env->f("acc_episodes"),
                       env->f("acc_steps"), c.seed, c.env_id_offset };
    return st;
}

int pp_step(ic3_env* env, const int32_t* actions, float* reward, int32_t* done, int32_t* alive, int32_t* is_completed,
            hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int G = group_lanes(env->dims.N);
    const long long threads = (long long)c.E * G;
    const int blocks = (int)((threads + 255) / 256);
    const StepOut out = { reward, done, alive, is_completed, env->d_err };
    hipLaunchKernelGGL(pp_step_kernel, dim3(blocks), dim3(256), 0, s, pp_state_of(env), out, actions, c.E, G);
    IC3_HIP(hipGetLastError());
    return 0;
}

int pp_observe(ic3_env* env, float* obs, hipStream_t s)
{
    const ic3_pp_cfg& c = env->pp;
    const int rows = env->dims.N;
    const int total = c.N + c.nprey, W = 2 * c.vision + 1, nseg = rows * W * W;
    const int vocab = c.dim * c.dim + 4;
    const size_t lds = (size_t)(((2 * total + 3) & ~3) + 2 * nseg) * sizeof(int32_t);
    // Geometry/stores chosen by measurement on MI355X (profiles/r01/obs_variants.txt, obs_geometry.txt): one WG per
    // env with LDS-staged descriptors, plain (not nontemporal) stores, 1 KiB-aligned wave stores; 1024 threads per env
    // (9 stores per thread for PP-hard) beat 256 (36 per thread) by 4 % — fewer stores per thread is better here.
    const long long Q = (long long)nseg * (vocab / 4);
    if ((vocab & 3) == 0 && Q >= 4096) {
        hipLaunchKernelGGL((pp_obs_kernel<true, false, true, 1024>), dim3(c.E), dim3(1024), lds, s, env->fv("loc_r"),
                           env->fv("loc_c"), obs, c.N, c.nprey, c.dim, c.vision, rows);
    } else if ((vocab & 3) == 0) {
        hipLaunchKernelGGL((pp_obs_kernel<true, false, true>), dim3(c.E), dim3(256), lds, s, env->fv("loc_r"),
                           env->fv("loc_c"), obs, c.N, c.nprey, c.dim, c.vision, rows);
    } else {   // vocab % 4 != 0 (odd dim): dword stores
        hipLaunchKernelGGL((pp_obs_kernel<false, false, false>), dim3(c.E), dim3(256), lds, s, env->fv("loc_r"),
                           env->fv("loc_c"), obs, c.N, c.nprey, c.dim, c.vision, rows);
    }
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // namespace ic3
