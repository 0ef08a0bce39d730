// env_device.hpp — device-side bodies of the Predator-Prey / Traffic-Junction step and of the sparse-encoder row,
// shared by the stand-alone kernels (pp_kernels.hip, tj_kernels.hip) and by the fused policy+step kernel
// (policy_step.hip), so that every launch geometry runs the same arithmetic.  gfx950 only.
//
// Reference semantics: /root/reference/ic3net-envs/ic3net_envs/predator_prey_env.py ("PP:line") and
// traffic_junction_env.py ("TJ:line").
#pragma once

#include <type_traits>

#include "ic3_common.hpp"

namespace ic3 {

typedef float dv_f32x4 __attribute__((ext_vector_type(4)));

// Sum over an aligned group of G lanes (G a power of two <= 64), result in every lane of the group.  The steps inside a
// 16-lane row are DPP moves on the VALU (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror) instead of ds_bpermute
// round trips; only the 16- and 32-lane steps go through __shfl_xor.
template <int G>
__device__ __forceinline__ float group_sum(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                     0xf, 0xf, false));
    };
    if constexpr (G >= 2) v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    if constexpr (G >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16);
    if constexpr (G >= 64) v += __shfl_xor(v, 32);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Predator-Prey
// ------------------------------------------------------------------------------------------------
// Auto-reset (ic3_env_set_auto_reset): an env whose episode ends at this step — episode_over, or max_steps steps played
// — starts its next episode inside the same launch, like the reference's loop `while ...: get_episode()`
// (trainer.py:107-108,227-242): episode += 1, t = 0, fresh state on the episode's own Philox key; `done` reports the
// end, and the finished episode's success flag / length go to per-env accumulators (env.stat is per episode).
struct AutoReset {
    int max_steps;                  // 0 = off (lock-step mode: finished envs freeze until reset())
    int32_t *episode;               // [E] episode counters (Philox key)
    int32_t *acc_success, *acc_episodes, *acc_steps;   // [E] sums over the episodes finished since reset()
    uint32_t seed, gid0;
};

struct PPState {
    int32_t *loc_r, *loc_c, *reached, *over, *success, *tstep;
    AutoReset ar;
    int Np;       // predators (cfg.N)
    int nprey;
    int dim, v, mode, naction;
    int rows;     // rows seen by the policy: Np, or Np + nprey with enemy_comm
};

struct StepOut {
    float* reward;
    int32_t *done, *alive_out, *comp_out, *err;
};

// x / d for 0 <= x < 2^20 through the float pipe: inv = 1.0f / d.  (x + 0.5) / d is at least 0.5 / d away from the next
// integer, the float error is below x * 2^-22 / d: exact.  A runtime integer division expands to ~25 VALU instructions.
__device__ __forceinline__ int div_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

__device__ __forceinline__ bool padded_outside(int pr, int pc, int v, int dim)
{
    return pr < v || pr >= v + dim || pc < v || pc >= v + dim;  // np.pad(grid, vision, OUTSIDE_CLASS) PP:184
}

// reset(): PP:146-175 = sequential rejection sampling of total = N + nprey DISTINCT cells on the injected stream
// (draw d of (episode ep, env gid): Philox domain DOMAIN_PP_RESET).  One lane per env — the draw sequence is serial.
// Shared by pp_reset_kernel and by the in-launch restart of pp_step_lanes (auto-reset).
__host__ __device__ inline void pp_place_entities(int32_t* rr, int32_t* cc, int total, int dim, uint32_t seed, uint32_t gid,
                                                  uint32_t ep)
{
    const uint32_t ncell = (uint32_t)(dim * dim);
    int m = 0;
    uint32_t d = 0;
    while (m < total) {
        const uint32_t k = scale24(philox_x24(seed, gid, DOMAIN_PP_RESET, ep, 0u, d), ncell);
        ++d;
        const int kr = (int)(k / (uint32_t)dim), kc = (int)(k % (uint32_t)dim);
        bool dup = false;
        for (int j = 0; j < m; ++j) dup |= (rr[j] == kr) & (cc[j] == kc);   // own earlier writes (same lane)
        if (!dup) {
            rr[m] = kr;
            cc[m] = kc;
            ++m;
        }
    }
}

// step: PP:112-144 = _take_action for every predator (PP:212-252), then _get_reward (PP:254-290).  Called by ALL
// lanes of a wave (ballots); lane (e, n) with n < G, G = pow2 >= rows lanes per env, groups aligned inside the wave.
// `act_of()` returns the env action of (e, n) — only evaluated for lanes with e < E and n < rows.
template <class ActFn>
__device__ __forceinline__ void pp_step_lanes(const PPState& s, const StepOut& o, int e, int n, int E, int G, ActFn act_of)
{
    const int N = s.Np, rows = s.rows, dim = s.dim, v = s.v, mode = s.mode;
    const bool in_env = (e < E) && (n < rows);
    const bool valid = in_env && (n < N);
    const int total = N + s.nprey;
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(G - 1);
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << gbase);

    int r = 0, c = 0, pr = -1, pc = -1, rch = 0, act = 4, was_over = 1, t_old = 0;
    if (in_env) {
        act = act_of();
        was_over = s.over[e];
        t_old = s.tstep[e];   // read by every lane in front of the ballots; lane 0 writes it behind them
        if (act > s.naction) atomicOr(o.err, 1);  // PP:137 (<=, quirk Q2; checked on every entry of `action`)
    }
    if (valid) {
        const size_t li = (size_t)e * total + n;
        r = s.loc_r[li];
        c = s.loc_c[li];
        pr = s.loc_r[(size_t)e * total + N];  // prey 0 only: (N,2)==(1,2) broadcast, quirk Q7 PP:258
        pc = s.loc_c[(size_t)e * total + N];
        rch = s.reached[(size_t)e * N + n];
    }
    const bool live = valid && !was_over;
    if (live && rch != 1 && act != 5) {  // frozen PP:221-222; (sic) STAY guard PP:224-226
        if (act == 0) {                  // UP PP:229-232
            int qr = r + v - 1;
            qr = qr < 0 ? 0 : qr;
            if (!padded_outside(qr, c + v, v, dim)) r = (r - 1 > 0) ? r - 1 : 0;
        } else if (act == 1) {           // RIGHT PP:235-239 (padded index clamped to dim-1, sic)
            int qc = c + v + 1;
            qc = qc > dim - 1 ? dim - 1 : qc;
            if (!padded_outside(r + v, qc, v, dim)) c = (c + 1 < dim - 1) ? c + 1 : dim - 1;
        } else if (act == 2) {           // DOWN PP:242-246
            int qr = r + v + 1;
            qr = qr > dim - 1 ? dim - 1 : qr;
            if (!padded_outside(qr, c + v, v, dim)) r = (r + 1 < dim - 1) ? r + 1 : dim - 1;
        } else if (act == 3) {           // LEFT PP:249-252
            int qc = c + v - 1;
            qc = qc < 0 ? 0 : qc;
            if (!padded_outside(r + v, qc, v, dim)) c = (c - 1 > 0) ? c - 1 : 0;
        }
    }
    const bool on = live && (r == pr) && (c == pc);
    const int n_on = __popcll(__ballot(on) & gmask);
    const int rch_new = (rch == 1 || on) ? 1 : 0;  // PP:271
    const int n_reached = __popcll(__ballot(live && rch_new) & gmask);
    if (!in_env) return;
    if (!valid) {   // prey row (enemy_comm): reward 0.05 while no predator is on it, else 0 (PP:276-281)
        o.reward[(size_t)e * rows + n] = was_over ? 0.0f : (n_on == 0 ? (float)0.05 : 0.0f);
        if (o.alive_out) o.alive_out[(size_t)e * rows + n] = 1;
        if (o.comp_out) o.comp_out[(size_t)e * rows + n] = 0;
        return;
    }
    // env-uniform: every lane of the group evaluates the same values
    const int ov_new = live ? ((n_reached == N && mode == IC3_PP_MIXED) ? 1 : 0) : was_over;   // PP:273-274
    const int t_new = t_old + (live ? 1 : 0);
    const bool restart = s.ar.max_steps > 0 && live && (ov_new || t_new >= s.ar.max_steps);
    float rew = 0.0f;
    if (live) {
        double rd = -0.05;  // TIMESTEP_PENALTY PP:256
        if (on) {
            if (mode == IC3_PP_COOPERATIVE) rd = 0.05 * (double)n_on;        // PP:262
            else if (mode == IC3_PP_COMPETITIVE) rd = 0.05 / (double)n_on;   // PP:265
            else rd = 0.0;                                                   // PP:267
        }
        rew = (float)rd;
        if (!restart) {                      // (a restarting env's state is written by lane 0 below)
            const size_t li = (size_t)e * total + n;
            s.loc_r[li] = r;
            s.loc_c[li] = c;
            s.reached[(size_t)e * N + n] = rch_new;
        }
    }
    o.reward[(size_t)e * rows + n] = rew;
    if (o.alive_out) o.alive_out[(size_t)e * rows + n] = 1;
    if (o.comp_out) o.comp_out[(size_t)e * rows + n] = 0;
    if (n == 0) {
        if (live && !restart) {
            if (mode != IC3_PP_COMPETITIVE) s.success[e] = (n_on == N) ? 1 : 0;    // PP:284-288
            s.over[e] = ov_new;
            s.tstep[e] = t_new;
        }
        o.done[e] = restart ? 1 : ov_new;
        if (restart) {
            // the finished episode's env.stat (PP:284-288) and length, then reset(): PP:146-175 on the next episode's key
            if (mode != IC3_PP_COMPETITIVE) s.ar.acc_success[e] += (n_on == N) ? 1 : 0;
            s.ar.acc_episodes[e] += 1;
            s.ar.acc_steps[e] += t_new;
            const uint32_t ep = (uint32_t)(s.ar.episode[e] + 1);
            pp_place_entities(s.loc_r + (size_t)e * total, s.loc_c + (size_t)e * total, total, dim, s.ar.seed,
                              s.ar.gid0 + (uint32_t)e, ep);
            for (int i = 0; i < N; ++i) s.reached[(size_t)e * N + i] = 0;
            s.over[e] = 0;
            s.success[e] = 0;
            s.ar.episode[e] = (int32_t)ep;
            s.tstep[e] = 0;
        }
    }
}

// Window descriptors of ONE env in LDS: sr/sc = positions of its Np + nprey entities, tab[a*W*W + dy*W + dx] =
// (one-hot channel, #predators | #prey << 16) for entity a's window cell (dy, dx).  Entry s of the env's table:
__device__ __forceinline__ int2 pp_tab_entry(const int32_t* sr, const int32_t* sc, int s, int N, int total, int dim, int v)
{
    const int W = 2 * v + 1;
    const int OUTSIDE = dim * dim + 1;
    const int a = div_small(s, 1.0f / (float)(W * W)), w = s - a * (W * W);
    const int dy = div_small(w, 1.0f / (float)W), dx = w - dy * W;
    const int gr = sr[a] + dy - v, gc = sc[a] + dx - v;
    const int id = (gr >= 0 && gr < dim && gc >= 0 && gc < dim) ? gr * dim + gc : OUTSIDE;
    int npred = 0, npr = 0;
    for (int p = 0; p < N; ++p) npred += (sr[p] == gr) & (sc[p] == gc);             // PP:191-192
    for (int p = N; p < total; ++p) npr += (sr[p] == gr) & (sc[p] == gc);           // PP:194-195
    return make_int2(id, npred | (npr << 16));
}

// The non-zero entries of one window cell of the observation (PP:177-210): `cell` = first float of the cell's `vocab`
// channels, d = its pp_tab_entry.  Channels: d.x one-hot (grid id or OUTSIDE), vocab-2 #prey, vocab-1 #predators (the
// counts ADD to the one-hot when they share a channel, quirk Q3).  The rest of the cell is zero.
__device__ __forceinline__ void pp_obs_patch(float* __restrict__ cell, int2 d, int vocab)
{
    const float npred = (float)(d.y & 0xffff), nprey = (float)(d.y >> 16);
    cell[d.x] = 1.f + (d.x == vocab - 2 ? nprey : 0.f) + (d.x == vocab - 1 ? npred : 0.f);
    if (d.x != vocab - 2 && nprey != 0.f) cell[vocab - 2] = nprey;
    if (d.x != vocab - 1 && npred != 0.f) cell[vocab - 1] = npred;
}

// A table of float4 rows (H4 float4s each) behind a plain pointer or a buffer descriptor (32-bit offsets on the lane,
// no 64-bit address arithmetic per gather).  `if (t)` = present.
struct PtrRows {
    const dv_f32x4* p;
    __device__ __forceinline__ dv_f32x4 at(int row, int H4, int c4) const { return p[(size_t)row * H4 + c4]; }
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};
struct BufRows {
    __amdgpu_buffer_rsrc_t r;
    bool present;
    __device__ __forceinline__ dv_f32x4 at(int row, int H4, int c4) const
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (row * H4 + c4) * 16, 0, 0);
        return __builtin_bit_cast(dv_f32x4, v);
    }
    __device__ __forceinline__ explicit operator bool() const { return present; }
};

// encoder(obs row of entity a)[4*c4 .. 4*c4+3] from the env's LDS descriptors (see pp_encode_kernel)
// `cells`: bit c set = window cell c may carry a predator / prey count (the centre is handled apart); ~0u = unknown.
// With the location table present only those cells are visited (in ascending order, as the full scan would): the
// fused policy+step kernel builds the bit set once per row next to the descriptors instead of letting each of the H/4
// lanes of a row re-scan all W*W cells.
template <class TW, class TL>
__device__ __forceinline__ dv_f32x4 pp_encode_row_t(const int32_t* sr, const int32_t* sc, const int2* tab, int a, int c4,
                                                    int H4, int WW, int vocab, int dim, const TW Wt,
                                                    const dv_f32x4* __restrict__ bias, const TL loc_table, unsigned cells)
{
    // The entity's own cell (the window centre) always carries a count: its two count rows are gathered
    // unconditionally (scaled by the counts; `+ 0 * w` is exact), together with the bias and the table row — four
    // independent loads in flight per row, and for most rows nothing else.  The other cells keep the data-dependent
    // gathers (a neighbour inside the window is the exception); making those unconditional too was measured slower.
    const int centre = WW >> 1;
    const int2 tc = tab[a * WW + centre];
    dv_f32x4 acc = bias[c4];
    const dv_f32x4 w_pred = Wt.at(centre * vocab + vocab - 1, H4, c4);
    const dv_f32x4 w_prey = Wt.at(centre * vocab + vocab - 2, H4, c4);
    // the one-hot location channels of all window cells depend only on the entity's position: one row of the
    // pre-summed table (pp_encode_table_kernel) replaces W*W gathered rows
    if (loc_table) acc += loc_table.at(sr[a] * dim + sc[a], H4, c4);
    acc += (float)(tc.y & 0xffff) * w_pred;
    acc += (float)(tc.y >> 16) * w_prey;
    if (loc_table && WW <= 32) {
        unsigned m = cells & ~(1u << centre) & (WW == 32 ? ~0u : ((1u << WW) - 1u));
        while (m) {
            const int cell = __builtin_ctz(m);
            m &= m - 1;
            const int2 t = tab[a * WW + cell];
            const int row = cell * vocab;
            const int npred = t.y & 0xffff, npr = t.y >> 16;
            if (npred) acc += (float)npred * Wt.at(row + vocab - 1, H4, c4);
            if (npr) acc += (float)npr * Wt.at(row + vocab - 2, H4, c4);
        }
        return acc;
    }
    for (int cell = 0; cell < WW; ++cell) {
        const int2 t = tab[a * WW + cell];
        const int row = cell * vocab;
        if (!loc_table) acc += Wt.at(row + t.x, H4, c4);
        if (cell == centre) continue;
        const int npred = t.y & 0xffff, npr = t.y >> 16;
        if (npred) acc += (float)npred * Wt.at(row + vocab - 1, H4, c4);
        if (npr) acc += (float)npr * Wt.at(row + vocab - 2, H4, c4);
    }
    return acc;
}

__device__ __forceinline__ dv_f32x4 pp_encode_row(const int32_t* sr, const int32_t* sc, const int2* tab, int a, int c4,
                                                  int H4, int WW, int vocab, int dim, const dv_f32x4* __restrict__ Wt,
                                                  const dv_f32x4* __restrict__ bias, const dv_f32x4* __restrict__ loc_table,
                                                  unsigned cells = ~0u)
{
    return pp_encode_row_t(sr, sc, tab, a, c4, H4, WW, vocab, dim, PtrRows{ Wt }, bias, PtrRows{ loc_table }, cells);
}

// ------------------------------------------------------------------------------------------------
// Traffic-Junction
// ------------------------------------------------------------------------------------------------
struct TJState {
    int32_t *alive, *wait, *loc_r, *loc_c, *last_act, *route_loc, *route_id, *completed, *cars, *failed;
    const int32_t *over, *episode;
    int32_t* tstep;
    AutoReset ar;
    const int32_t *route_off, *route_rc, *grid, *thr;
    int N, narrival, rpa;
    int h, w, v, vocab, outside, car_class, npath, hdr;
    uint32_t seed, gid0;
};

// reset() of one car slot (TJ:160-190); shared by tj_reset_kernel and the in-launch restart of tj_step_lanes
__host__ __device__ inline void tj_reset_car(int32_t* alive, int32_t* wait, int32_t* loc_r, int32_t* loc_c, int32_t* last_act,
                                             int32_t* route_loc, int32_t* route_id, int32_t* completed, size_t i)
{
    alive[i] = 0;       // TJ:171
    wait[i] = 0;        // TJ:172
    loc_r[i] = 0;       // TJ:187
    loc_c[i] = 0;
    last_act[i] = 0;    // TJ:188
    route_loc[i] = -1;  // TJ:190
    route_id[i] = -1;   // TJ:178
    completed[i] = 0;
}

// step: TJ:206-252 = _take_action (TJ:540-581), _add_cars (TJ:369-393), _get_reward (TJ:585-595).  Called by all lanes
// of a wave; G = pow2 >= max(N, 8) lanes per env.
template <class ActFn>
__device__ __forceinline__ void tj_step_lanes(const TJState& s, const StepOut& o, int e, int n, int E, int G, ActFn act_of)
{
    const int N = s.N, narrival = s.narrival, rpa = s.rpa;
    const bool env_ok = e < E;
    const bool valid = env_ok && n < N;
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(G - 1);
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << gbase);
    const size_t i = (size_t)e * N + n;

    int alive = 0, wait = 0, r = 0, c = 0, last_act = 0, rloc = -1, rid = -1, completed = 0, act = 1;
    uint32_t ep = 0, t = 0;
    if (env_ok) {
        ep = (uint32_t)s.episode[e];
        t = (uint32_t)s.tstep[e];
    }
    if (valid) {
        alive = s.alive[i];
        wait = s.wait[i];
        r = s.loc_r[i];
        c = s.loc_c[i];
        last_act = s.last_act[i];
        rloc = s.route_loc[i];
        rid = s.route_id[i];
        act = act_of();
        if (act > 2) atomicOr(o.err, 1);  // TJ:228 (naction = 2, <=, quirk Q2)
    }
    // ---- _take_action TJ:540-581 ----
    if (valid && alive) {
        wait += 1;                      // TJ:546
        if (act == 1) {
            last_act = 1;               // TJ:549-551
        } else if (act == 0) {
            rloc += 1;                  // TJ:556
            const int off = s.route_off[rid], len = s.route_off[rid + 1] - off;
            if (rloc == len) {          // TJ:560-568 reached the end of its route
                alive = 0;
                wait = 0;
                r = c = 0;
                completed = 1;
            } else {
                const int rc = s.route_rc[off + rloc];  // TJ:575-578
                r = rc >> 16;
                c = rc & 0xffff;
                last_act = 0;           // TJ:581
            }
        }
    }
    // ---- _add_cars TJ:369-393: lane j of the group pre-draws arrival point j's three uniforms ----
    const int32_t thr = *s.thr;  // floor(add_rate * 2^24): u <= add_rate <=> x24 <= thr (exact)
    uint32_t x0 = 0, x1 = 0, x2 = 0;
    if (env_ok && n < narrival) {
        x0 = philox_x24(s.seed, s.gid0 + (uint32_t)e, DOMAIN_TJ_ADD, ep, t, 3u * n + 0u);
        x1 = philox_x24(s.seed, s.gid0 + (uint32_t)e, DOMAIN_TJ_ADD, ep, t, 3u * n + 1u);
        x2 = philox_x24(s.seed, s.gid0 + (uint32_t)e, DOMAIN_TJ_ADD, ep, t, 3u * n + 2u);
    }
    for (int a = 0; a < narrival; ++a) {
        const unsigned long long am = __ballot(valid && alive) & gmask;
        const unsigned long long dm = __ballot(valid && !alive) & gmask;
        const int cars = __popcll(am), nd = __popcll(dm);
        const uint32_t u0 = (uint32_t)__shfl((int)x0, gbase + a);
        const uint32_t u1 = (uint32_t)__shfl((int)x1, gbase + a);
        const uint32_t u2 = (uint32_t)__shfl((int)x2, gbase + a);
        const bool add = (cars < N) && ((int32_t)u0 <= thr);              // TJ:371-372, 375
        if (add && valid && !alive) {
            const int k = (int)scale24(u1, (uint32_t)nd);                 // _choose_dead TJ:614-618: k-th dead slot
            const int rank = __popcll(dm & ((1ull << lane) - 1ull));
            if (rank == k) {
                alive = 1;                                                // TJ:380
                rid = (int)scale24(u2, (uint32_t)rpa) + a * rpa;          // TJ:383-385
                rloc = 0;                                                 // TJ:389
                const int rc = s.route_rc[s.route_off[rid]];              // TJ:390
                r = rc >> 16;
                c = rc & 0xffff;
            }
        }
    }
    const int cars_now = __popcll(__ballot(valid && alive) & gmask);     // == cars_in_sys (TJ:393,561)
    // ---- _get_reward TJ:585-595: crash iff another car (alive or parked dead) shares a non-(0,0) cell ----
    const int packed = valid ? ((r << 16) | c) : -1;
    bool same = false;
    for (int j = 0; j < N; ++j) {
        const int pj = __shfl(packed, gbase + j);
        same |= (j != n) && (pj == packed);
    }
    const bool crash = valid && same && (packed != 0);                   // l.any(): loc != (0,0), quirk Q10
    const bool any_crash = (__ballot(crash) & gmask) != 0ull;
    if (!valid) return;
    double rd = -0.01 * (double)wait;                                     // TJ:586
    if (crash) rd += -10.0;                                               // TJ:591
    rd = (double)alive * rd;                                              // TJ:594
    o.reward[i] = (float)rd;
    if (o.alive_out) o.alive_out[i] = alive;                              // info['alive_mask'] TJ:244
    if (o.comp_out) o.comp_out[i] = completed;                            // info['is_completed'] TJ:247
    const bool restart = s.ar.max_steps > 0 && (int)t + 1 >= s.ar.max_steps;   // TJ episodes always run max_steps (Q12)
    if (!restart) {
        s.alive[i] = alive;
        s.wait[i] = wait;
        s.loc_r[i] = r;
        s.loc_c[i] = c;
        s.last_act[i] = last_act;
        s.route_loc[i] = rloc;
        s.route_id[i] = rid;
        s.completed[i] = completed;
    } else {                                                              // reset(): TJ:160-190
        tj_reset_car(s.alive, s.wait, s.loc_r, s.loc_c, s.last_act, s.route_loc, s.route_id, s.completed, i);
    }
    if (n == 0) {
        if (!restart) {
            s.cars[e] = cars_now;
            if (any_crash) s.failed[e] = 1;                               // TJ:592
            s.tstep[e] = (int32_t)t + 1;
            o.done[e] = s.over[e];                                        // never set by TJ (quirk Q12)
        } else {
            const int failed = (any_crash || s.failed[e]) ? 1 : 0;
            s.ar.acc_success[e] += 1 - failed;                            // stat['success'] = 1 - has_failed, TJ:249
            s.ar.acc_episodes[e] += 1;
            s.ar.acc_steps[e] += (int32_t)t + 1;
            s.cars[e] = 0;
            s.failed[e] = 0;
            s.ar.episode[e] = (int32_t)ep + 1;
            s.tstep[e] = 0;
            o.done[e] = 1;
        }
    }
}

// Per-env LDS block of the TJ observation / encoder kernels: [sr | sc | sal | s0 | s1 | s2 | s3] (N words each), then
// tab[N*WW] = (one-hot channel, #cars).
struct TJTile {
    int32_t *sr, *sc, *sal;
    float *s0, *s1, *s2, *s3;
    int2* tab;
};

__device__ __forceinline__ int tj_tile_words(int N, int WW) { return ((7 * N + 3) & ~3) + 2 * N * WW; }

__device__ __forceinline__ TJTile tj_tile_at(int32_t* base, int N)
{
    TJTile t;
    t.sr = base;
    t.sc = t.sr + N;
    t.sal = t.sc + N;
    t.s0 = reinterpret_cast<float*>(t.sal + N);
    t.s1 = t.s0 + N;
    t.s2 = t.s1 + N;
    t.s3 = t.s2 + N;
    t.tab = reinterpret_cast<int2*>(base + ((7 * N + 3) & ~3));
    return t;
}

__device__ __forceinline__ void tj_tile_load_car(const TJTile& t, const TJState& s, int e, int a)
{
    const size_t i = (size_t)e * s.N + a;
    t.sr[a] = s.loc_r[i];
    t.sc[a] = s.loc_c[i];
    t.sal[a] = s.alive[i];
    t.s0[a] = (float)((double)s.last_act[i] / 1.0);                          // TJ:338 naction-1 == 1
    t.s1[a] = (float)((double)s.route_id[i] / (double)(s.npath - 1));        // TJ:341
    t.s2[a] = (float)((double)t.sr[a] / (double)(s.h - 1));                  // TJ:344
    t.s3[a] = (float)((double)t.sc[a] / (double)(s.w - 1));
}

__device__ __forceinline__ int2 tj_tab_entry(const TJTile& t, const TJState& s, int q)
{
    const int W = 2 * s.v + 1, WW = W * W;
    const int a = div_small(q, 1.0f / (float)WW), cell = q - a * WW;
    const int dy = div_small(cell, 1.0f / (float)W), dx = cell - dy * W;
    const int gr = t.sr[a] + dy - s.v, gc = t.sc[a] + dx - s.v;
    const int id = (gr >= 0 && gr < s.h && gc >= 0 && gc < s.w) ? s.grid[gr * s.w + gc] : s.outside;   // pad_grid TJ:317
    int ncar = 0;
    for (int p = 0; p < s.N; ++p) ncar += (t.sr[p] == gr) & (t.sc[p] == gc);                          // TJ:326-327
    return make_int2(id, ncar);
}

// The non-zero entries of ONE env's observation rows (TJ:331-344,352-356) on top of a zero-filled chunk: item q < N is
// the header of car q's row, item N + a * WW + cell one window cell of car a's row (<= 2 entries: the one-hot id and the
// car count).  Rows of dead cars stay zero.  `env_rows` = first float of the env's N rows.
__device__ __forceinline__ void tj_obs_patch(const TJTile& t, const TJState& s, float* __restrict__ env_rows, int obs_dim,
                                             int WW, int q)
{
    const int N = s.N;
    if (q < N) {                                     // header of car q's row (TJ:338-344)
        if (!t.sal[q]) return;
        float* row = env_rows + (size_t)q * obs_dim;
        row[0] = t.s0[q];
        row[1] = t.s1[q];
        if (s.hdr == 4) {
            row[2] = t.s2[q];
            row[3] = t.s3[q];
        }
    } else {                                         // window cell (TJ:331-332,352-356)
        const int qq = q - N, car = div_small(qq, 1.0f / (float)WW), cellx = qq - car * WW;
        if (!t.sal[car]) return;
        const int2 d = t.tab[qq];
        float* cell = env_rows + (size_t)car * obs_dim + s.hdr + (size_t)cellx * s.vocab;
        if (d.x >= 0) cell[d.x] = 1.f + (d.x == s.car_class ? (float)d.y : 0.f);
        if (d.x != s.car_class && d.y != 0) cell[s.car_class] = (float)d.y;
    }
}

// encoder(obs row of car a)[4*c4 ..] (see tj_encode_kernel): bias only for a dead car (its obs row is zero)
template <class TW, class TL>
__device__ __forceinline__ dv_f32x4 tj_encode_row_t(const TJTile& t, const TJState& s, int a, int c4, int H4, const TW Wt,
                                                    const dv_f32x4* __restrict__ bias, const TL loc_table, unsigned cells)
{
    const int W = 2 * s.v + 1, WW = W * W, centre = WW >> 1;
    dv_f32x4 acc = bias[c4];
    if (t.sal[a]) {
        // header rows, the table row and the car-count row of the car's own cell (count >= 1): independent loads
        const dv_f32x4 w0 = Wt.at(0, H4, c4), w1 = Wt.at(1, H4, c4);
        const dv_f32x4 w_car = Wt.at(s.hdr + centre * s.vocab + s.car_class, H4, c4);
        acc += t.s0[a] * w0;
        acc += t.s1[a] * w1;
        if (s.hdr == 4) {
            acc += t.s2[a] * Wt.at(2, H4, c4);
            acc += t.s3[a] * Wt.at(3, H4, c4);
        }
        if (loc_table) acc += loc_table.at(t.sr[a] * s.w + t.sc[a], H4, c4);   // see pp_encode_kernel
        acc += (float)t.tab[a * WW + centre].y * w_car;
        if (loc_table && WW <= 32) {               // only the cells that carry a car count (see pp_encode_row)
            unsigned m = cells & ~(1u << centre) & (WW == 32 ? ~0u : ((1u << WW) - 1u));
            while (m) {
                const int cell = __builtin_ctz(m);
                m &= m - 1;
                const int2 d = t.tab[a * WW + cell];
                if (d.y) acc += (float)d.y * Wt.at(s.hdr + cell * s.vocab + s.car_class, H4, c4);
            }
            return acc;
        }
        for (int cell = 0; cell < WW; ++cell) {
            const int2 d = t.tab[a * WW + cell];
            const int row = s.hdr + cell * s.vocab;
            if (!loc_table && d.x >= 0) acc += Wt.at(row + d.x, H4, c4);   // scalar vocab: -1 = not a road cell
            if (cell != centre && d.y) acc += (float)d.y * Wt.at(row + s.car_class, H4, c4);
        }
    }
    return acc;
}

__device__ __forceinline__ dv_f32x4 tj_encode_row(const TJTile& t, const TJState& s, int a, int c4, int H4,
                                                  const dv_f32x4* __restrict__ Wt, const dv_f32x4* __restrict__ bias,
                                                  const dv_f32x4* __restrict__ loc_table, unsigned cells = ~0u)
{
    return tj_encode_row_t(t, s, a, c4, H4, PtrRows{ Wt }, bias, PtrRows{ loc_table }, cells);
}

// ------------------------------------------------------------------------------------------------
// Observation rows of a RUN of consecutive envs [e0, e0 + nenv) streamed out by a group of `nthr` threads (thread
// index `t`), restricted to the slice [part, part + 1) / nparts of the run, so that a caller can issue the stores in
// pieces between other work.  Same element values as pp_obs_kernel / tj_obs_kernel (predator_prey_env.py:188-210,
// traffic_junction_env.py:321-366); descriptors in LDS: PP tab[el*N*WW + a*WW + cell], TJ one TJTile per env.
// The runs of consecutive envs are contiguous in the obs tensor: one address range per call.
// ------------------------------------------------------------------------------------------------
// PP, vocab % 4 == 0: every float4 lies inside one window cell; lane -> float4 map shifted so that wave stores are
// 1 KiB-aligned in the global address space (see pp_obs_kernel).
// NTS: non-temporal stores (the caller streams other data through the L2 next to the rows — commnet_fwd.hip)
template <bool NTS = false>
__device__ __forceinline__ void pp_obs_store_run(const int2* tab, float* __restrict__ obs, long long e0, int nenv,
                                                 int nsegE, int vocab, int t, int nthr, int part, int nparts)
{
    const int segq = vocab >> 2;
    const long long Qe = (long long)nsegE * segq;                 // float4s per env
    const int Q = (int)(Qe * nenv);                               // float4s of the run
    dv_f32x4* out = reinterpret_cast<dv_f32x4*>(obs) + e0 * Qe;
    const int o = (int)((e0 * Qe) & 63);
    const int rounds = (Q + o + nthr - 1) / nthr;                 // passes of the whole group over the (shifted) run
    const int r0 = (int)((long long)rounds * part / nparts), r1 = (int)((long long)rounds * (part + 1) / nparts);
    const float inv_segq = 1.0f / (float)segq;
    const bool small = Q < (1 << 20);                             // (g + 0.5) / segq is exact in fp32 below 2^20
    for (int r = r0; r < r1; ++r) {
        const int g = r * nthr + t - o;
        if (g < 0 || g >= Q) continue;
        const int seg = small ? (int)(((float)g + 0.5f) * inv_segq) : g / segq;
        const int q = g - seg * segq;
        const int2 d = tab[seg];
        dv_f32x4 z = { 0.f, 0.f, 0.f, 0.f };
        if ((d.x >> 2) == q) {
            const int j = d.x & 3;
            z.x = (j == 0) ? 1.f : 0.f;
            z.y = (j == 1) ? 1.f : 0.f;
            z.z = (j == 2) ? 1.f : 0.f;
            z.w = (j == 3) ? 1.f : 0.f;
        }
        if (q == segq - 1) {
            z.z += (float)(d.y >> 16);
            z.w += (float)(d.y & 0xffff);
        }
        if constexpr (NTS) __builtin_nontemporal_store(z, out + g);
        else out[g] = z;
    }
}

// PP, any vocab: dword stores
__device__ __forceinline__ void pp_obs_store_run_scalar(const int2* tab, float* __restrict__ obs, long long e0, int nenv,
                                                        int nsegE, int vocab, int t, int nthr, int part, int nparts)
{
    const long long Le = (long long)nsegE * vocab;
    const int L = (int)(Le * nenv);
    float* out = obs + e0 * Le;
    const int rounds = (L + nthr - 1) / nthr;
    const int r0 = (int)((long long)rounds * part / nparts), r1 = (int)((long long)rounds * (part + 1) / nparts);
    for (int r = r0; r < r1; ++r) {
        const int g = r * nthr + t;
        if (g >= L) continue;
        const int seg = g / vocab, ch = g - seg * vocab;
        const int2 d = tab[seg];
        float z = (ch == d.x) ? 1.f : 0.f;
        if (ch == vocab - 2) z += (float)(d.y >> 16);
        if (ch == vocab - 1) z += (float)(d.y & 0xffff);
        out[g] = z;
    }
}

// TJ element `off` of car a's row (TJ:336-362) from the env's LDS tile
__device__ __forceinline__ float tj_obs_value(const TJTile& t, const TJState& s, int a, int off, int WW, float inv_vocab)
{
    if (!t.sal[a]) return 0.0f;
    if (off < s.hdr) return off == 0 ? t.s0[a] : off == 1 ? t.s1[a] : off == 2 ? t.s2[a] : t.s3[a];
    const int k = off - s.hdr;
    const int seg = (int)(((float)k + 0.5f) * inv_vocab);        // exact for k < 2^20
    const int ch = k - seg * s.vocab;
    const int2 d = t.tab[a * WW + seg];
    float z = (ch == d.x) ? 1.0f : 0.0f;
    if (ch == s.car_class) z += (float)d.y;
    return z;
}

// TJ: rows are hdr + WW*vocab floats (no 16-byte structure): <= 3 head and tail dwords, float4 body with 1 KiB-aligned
// wave stores, every element evaluated on its own (see tj_obs_vec4_kernel).  tiles: LDS base of the run's TJTiles.
__device__ __forceinline__ void tj_obs_store_run(int32_t* tiles, int tile_words, const TJState& s, float* __restrict__ obs,
                                                 long long e0, int nenv, int t, int nthr, int part, int nparts)
{
    const int N = s.N, W = 2 * s.v + 1, WW = W * W, obs_dim = s.hdr + WW * s.vocab;
    const float inv_vocab = 1.0f / (float)s.vocab;
    const int Le = N * obs_dim;
    const int L = Le * nenv;
    const long long b0 = e0 * (long long)Le;
    float* out = obs + b0;
    const float inv_row = 1.0f / (float)obs_dim;
    const bool small = L < (1 << 20);                      // (f + 0.5) / obs_dim is exact in fp32 below 2^20
    auto value = [&](int f) -> float {
        const int row = small ? (int)(((float)f + 0.5f) * inv_row) : f / obs_dim;   // row of the run = el * N + a
        const int off = f - row * obs_dim;
        const int el = row / N, a = row - el * N;          // N <= 64: cheap
        return tj_obs_value(tj_tile_at(tiles + el * tile_words, N), s, a, off, WW, inv_vocab);
    };
    const int head = (int)((4 - (b0 & 3)) & 3);
    const int nb = (L - head) >> 2, tail = (L - head) & 3;
    if (part == 0) {
        if (t < head && t < L) out[t] = value(t);
        if (t < tail) {
            const int f = head + 4 * nb + t;
            out[f] = value(f);
        }
    }
    dv_f32x4* out4 = reinterpret_cast<dv_f32x4*>(out + head);
    const int o = (int)(((b0 + head) >> 2) & 63);
    const int rounds = (nb + o + nthr - 1) / nthr;
    const int r0 = (int)((long long)rounds * part / nparts), r1 = (int)((long long)rounds * (part + 1) / nparts);
    for (int r = r0; r < r1; ++r) {
        const int j = r * nthr + t - o;
        if (j < 0 || j >= nb) continue;
        const int f = head + 4 * j;
        dv_f32x4 z;
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = value(f + i);
        out4[j] = z;
    }
}

// host helpers (pp_kernels.hip / tj_kernels.hip): device views of a handle's state
PPState pp_state_of(const ic3_env* env);
TJState tj_state_of(const ic3_env* env);
int tj_group(int N);

}  // namespace ic3
