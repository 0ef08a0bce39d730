/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the ic3net_amd rollout engine.
 *
 * Plain-C restatement of the reference's per-environment algorithms (one env object at a time, the
 * way the reference runs them), used ONLY by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / reported baseline.  The product path (ic3net_amd/) never links,
 * imports or calls anything in this directory.
 *
 * Parity status: PINNED — every function below is checked against golden vectors captured from the
 * reference itself running in the build container (tests/golden/make_golden.py, fixtures
 * tests/golden/ (npz); tests/test_oracle_golden.py).
 *
 * Reference citations are relative to /root/reference.
 *   PP  = ic3net-envs/ic3net_envs/predator_prey_env.py
 *   TJ  = ic3net-envs/ic3net_envs/traffic_junction_env.py
 * The random source is the injected counter-based stream described in oracle/philox.py.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------- */
/* Philox4x32-10 (Random123); contract in oracle/philox.py                                       */
/* ------------------------------------------------------------------------------------------- */
static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { DOMAIN_PP_RESET = 1, DOMAIN_TJ_ADD = 2, DOMAIN_SAMPLE = 3, DOMAIN_BENCH = 4 };

uint32_t orc_x24(uint32_t seed, uint32_t env_gid, uint32_t domain, uint32_t episode, uint32_t t,
                 uint32_t draw)
{
    uint32_t ctr[4] = { draw, t, episode, domain }, key[2] = { seed, env_gid }, out[4];
    philox4x32_10(ctr, key, out);
    return out[0] >> 8;
}

void orc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

/* ------------------------------------------------------------------------------------------- */
/* Predator-Prey                                                                               */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N;        /* npredator = args.nfriendly            PP:80  */
    int32_t nprey;    /* args.nenemies (only 1 works, quirk Q7) PP:79  */
    int32_t dim;      /* PP:81 */
    int32_t vision;   /* PP:75 */
    int32_t mode;     /* 0 mixed, 1 cooperative, 2 competitive  PP:261-269 */
    int32_t naction;  /* 5, or 4 with --no_stay                 PP:90-93 */
    int32_t enemy_comm; /* prey rows in obs / reward (PP:203-207,255,276-281); the policy then sees N+nprey agents */
} orc_pp_cfg;

/* PP:146-168 reset + PP:173-175 _get_cordinates with the injected stream:
 * np.random.choice(dim*dim, N+nprey, replace=False) == sequential rejection sampling of distinct
 * cells, draw index 0,1,2,...; loc = unravel_index(cell, (dim,dim)) = (cell / dim, cell % dim).
 * loc layout: [(N+nprey)][2] (row, col), predators first (PP:159). */
void orc_pp_reset(const orc_pp_cfg* c, uint32_t seed, uint32_t env_gid, uint32_t episode,
                  int32_t* loc, int32_t* reached, int32_t* episode_over)
{
    const int total = c->N + c->nprey;
    const uint32_t ncell = (uint32_t)(c->dim * c->dim);
    int32_t chosen[1024];
    int n = 0;
    uint32_t d = 0;
    while (n < total) {
        uint32_t k = (uint32_t)(((uint64_t)orc_x24(seed, env_gid, DOMAIN_PP_RESET, episode, 0, d) * ncell) >> 24);
        ++d;
        int dup = 0;
        for (int j = 0; j < n; ++j) dup |= (chosen[j] == (int32_t)k);
        if (!dup) chosen[n++] = (int32_t)k;
    }
    for (int i = 0; i < total; ++i) {
        loc[2 * i + 0] = chosen[i] / c->dim;
        loc[2 * i + 1] = chosen[i] % c->dim;
    }
    memset(reached, 0, sizeof(int32_t) * (size_t)c->N);   /* PP:155 */
    *episode_over = 0;                                     /* PP:154 */
}

/* PP:212-252 _take_action for one predator.  The grid lookups in the reference test the padded id
 * grid for OUTSIDE_CLASS after clamping the index, which (probed, SURVEY B.5 iii) is a clamp-move:
 * we restate the actual lookups to keep vision==0 (no padding) identical. */
static void pp_take_action(const orc_pp_cfg* c, int32_t* loc, const int32_t* reached, int idx, int act)
{
    if (idx >= c->N) return;                 /* fixed prey                PP:214-219 */
    if (reached[idx] == 1) return;           /* frozen                    PP:221-222 */
    if (act == 5) return;                    /* (sic) STAY guard, Q1      PP:224-226 */
    const int v = c->vision, dim = c->dim;
    int32_t* r = &loc[2 * idx + 0];
    int32_t* q = &loc[2 * idx + 1];
    /* padded grid value at (pr, pc): OUTSIDE iff outside [v, v+dim) in either coordinate (PP:184) */
#define PADDED_OUTSIDE(pr, pc) ((pr) < v || (pr) >= v + dim || (pc) < v || (pc) >= v + dim)
    if (act == 0) {                                              /* UP    PP:229-232 */
        int pr = *r + v - 1; if (pr < 0) pr = 0;
        if (!PADDED_OUTSIDE(pr, *q + v)) *r = (*r - 1 > 0) ? *r - 1 : 0;
    } else if (act == 1) {                                       /* RIGHT PP:235-239 */
        int pc = *q + v + 1; if (pc > dim - 1) pc = dim - 1;     /* (sic) clamps padded idx to dim-1 */
        if (!PADDED_OUTSIDE(*r + v, pc)) *q = (*q + 1 < dim - 1) ? *q + 1 : dim - 1;
    } else if (act == 2) {                                       /* DOWN  PP:242-246 */
        int pr = *r + v + 1; if (pr > dim - 1) pr = dim - 1;
        if (!PADDED_OUTSIDE(pr, *q + v)) *r = (*r + 1 < dim - 1) ? *r + 1 : dim - 1;
    } else if (act == 3) {                                       /* LEFT  PP:249-252 */
        int pc = *q + v - 1; if (pc < 0) pc = 0;
        if (!PADDED_OUTSIDE(*r + v, pc)) *q = (*q - 1 > 0) ? *q - 1 : 0;
    }
#undef PADDED_OUTSIDE
}

/* PP:254-290 _get_reward (enemy_comm=False).  reward is float64 like the reference. */
static void pp_get_reward(const orc_pp_cfg* c, const int32_t* loc, int32_t* reached, double* reward,
                          int32_t* episode_over, int32_t* success)
{
    const int N = c->N;
    int n_on = 0;
    /* PP:258: np.all(predator_loc == prey_loc, axis=1) broadcasts (N,2)==(1,2): prey 0 only (Q7) */
    const int32_t pr = loc[2 * N + 0], pc = loc[2 * N + 1];
    for (int i = 0; i < N; ++i) n_on += (loc[2 * i] == pr && loc[2 * i + 1] == pc);
    for (int i = 0; i < N; ++i) {
        const int on = (loc[2 * i] == pr && loc[2 * i + 1] == pc);
        double r = -0.05;                                                 /* PP:256 */
        if (on) {
            if (c->mode == 1) r = 0.05 * (double)n_on;                    /* PP:262 */
            else if (c->mode == 2) r = 0.05 / (double)n_on;               /* PP:265 */
            else r = 0.0;                                                 /* PP:267 */
            reached[i] = 1;                                               /* PP:271 */
        }
        reward[i] = r;
    }
    int all = 1;
    for (int i = 0; i < N; ++i) all &= (reached[i] == 1);
    if (all && c->mode == 0) *episode_over = 1;                           /* PP:273-274 */
    if (c->enemy_comm)                                                    /* PP:255,276-281 prey reward */
        for (int p = 0; p < c->nprey; ++p) reward[N + p] = (n_on == 0) ? 0.05 : 0.0;
    if (c->mode != 2) *success = (n_on == N) ? 1 : 0;                     /* PP:284-288 */
}

/* PP:112-144 step.  Returns 0, -1 if the episode is already over (RuntimeError), -2 on a bad action
 * (the reference asserts AFTER moving, PP:137; we report it and leave state moved likewise). */
int orc_pp_step(const orc_pp_cfg* c, const int32_t* action, int32_t* loc, int32_t* reached,
                double* reward, int32_t* episode_over, int32_t* success)
{
    if (*episode_over) return -1;                                         /* PP:129-130 */
    int bad = 0;
    const int rows = c->N + (c->enemy_comm ? c->nprey : 0);               /* len(action) == args.nagents */
    for (int i = 0; i < rows; ++i) {                                      /* PP:134-135 (prey actions are ignored) */
        pp_take_action(c, loc, reached, i, action[i]);
        bad |= (action[i] > c->naction);                                  /* PP:137 (<=, Q2) */
    }
    if (bad) return -2;
    *episode_over = 0;                                                    /* PP:140 */
    /* obs is produced by orc_pp_obs BEFORE the reward pass freezes anything (PP:141-144) */
    pp_get_reward(c, loc, reached, reward, episode_over, success);
    return 0;
}

/* PP:188-210 _get_obs + PP:177-186 _set_grid + env_wrappers.py:88-100 _flatten_obs.
 * Dense rows: obs[a][(dy*W+dx)*vocab + ch], W = 2v+1, vocab = dim*dim+4.
 * The reference copies a (dim+2v)^2 x vocab one-hot base grid, adds 1 at PREDATOR_CLASS for every
 * predator and at PREY_CLASS for every prey (counts, Q3), then slices each predator's window. */
void orc_pp_obs(const orc_pp_cfg* c, const int32_t* loc, float* obs)
{
    const int N = c->N, v = c->vision, dim = c->dim, W = 2 * v + 1;
    const int base = dim * dim, OUTSIDE = base + 1, PREY = base + 2, PRED = base + 3, vocab = base + 4;
    const int pd = dim + 2 * v;
    const int rows = N + (c->enemy_comm ? c->nprey : 0);                  /* PP:203-207 */
    memset(obs, 0, sizeof(float) * (size_t)rows * W * W * vocab);
    for (int a = 0; a < rows; ++a) {
        float* row = obs + (size_t)a * W * W * vocab;
        for (int dy = 0; dy < W; ++dy)
            for (int dx = 0; dx < W; ++dx) {
                const int pr = loc[2 * a] + dy, pc = loc[2 * a + 1] + dx;    /* padded coords PP:199-201 */
                (void)pd;
                float* cell = row + (size_t)(dy * W + dx) * vocab;
                const int gr = pr - v, gc = pc - v;
                const int id = (gr >= 0 && gr < dim && gc >= 0 && gc < dim) ? gr * dim + gc : OUTSIDE; /* PP:178,184 */
                cell[id] += 1.0f;                                             /* PP:186 one-hot */
                for (int p = 0; p < N; ++p)                                   /* PP:191-192 */
                    if (loc[2 * p] + v == pr && loc[2 * p + 1] + v == pc) cell[PRED] += 1.0f;
                for (int p = 0; p < c->nprey; ++p)                            /* PP:194-195 */
                    if (loc[2 * (N + p)] + v == pr && loc[2 * (N + p) + 1] + v == pc) cell[PREY] += 1.0f;
            }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Traffic-Junction                                                                            */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N;            /* ncar = args.nagents                      TJ:88  */
    int32_t h, w;         /* self.dims (dim+1 for easy)               TJ:111-115 */
    int32_t vision;
    int32_t vocab;        /* BASE + 3                                 TJ:134 */
    int32_t outside;      /* OUTSIDE_CLASS = BASE                     TJ:131 */
    int32_t car_class;    /* CAR_CLASS = BASE + 2                     TJ:132 */
    int32_t npath;        /* nPr(nroad, 2)                            TJ:126 */
    int32_t narrival;     /* len(self.routes)                         TJ:370 */
    int32_t routes_per_arrival; /* len(routes) for every arrival point TJ:384 */
    int32_t scalar;       /* vocab_type == 'scalar' (TJ:139-148): grid holds road flags, obs rows are
                             [last_act, route, r/(h-1), c/(w-1), (road, #cars) per window cell] */
} orc_tj_cfg;

/* tables: grid[h*w] road ids (TJ:300-319); route_off[npath+1]; route_rc[2*total_cells] (row,col) */

/* TJ:160-204 reset (state part; curriculum is a host-side scalar, see orc_tj_curriculum) */
void orc_tj_reset(const orc_tj_cfg* c, int32_t* alive, int32_t* wait, int32_t* loc, int32_t* last_act,
                  int32_t* route_loc, int32_t* route_id, int32_t* cars_in_sys, int32_t* has_failed)
{
    for (int i = 0; i < c->N; ++i) {
        alive[i] = 0; wait[i] = 0;                    /* TJ:171-172 */
        route_id[i] = -1;                             /* TJ:178 */
        loc[2 * i] = loc[2 * i + 1] = 0;              /* TJ:187 */
        last_act[i] = 0;                              /* TJ:188 */
        route_loc[i] = -1;                            /* TJ:190 */
    }
    *cars_in_sys = 0;                                 /* TJ:173 */
    *has_failed = 0;                                  /* TJ:169 */
}

/* Python's float `//` (CPython float_divmod): NOT floor(a/b) — e.g. 0.12 // 0.01 == 11.0 because
 * fmod(0.12, 0.01) = 0.00999.. ; the reference's add_rate = 0.01 * (exact_rate // 0.01) (TJ:626, Q16). */
static double py_float_floordiv(double vx, double wx)
{
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) { mod += wx; div -= 1.0; }
    }
    double fl;
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else {
        fl = copysign(0.0, vx / wx);
    }
    return fl;
}

/* TJ:620-626 curriculum + TJ:196-200 gating; state = (exact_rate, add_rate, epoch_last_update) */
void orc_tj_curriculum(double add_rate_min, double add_rate_max, double curr_start, double curr_end,
                       int has_epoch, double epoch, double* exact_rate, double* add_rate,
                       double* epoch_last_update)
{
    const double epoch_range = curr_end - curr_start, add_rate_range = add_rate_max - add_rate_min;
    if (has_epoch && epoch_range > 0 && add_rate_range > 0 && epoch > *epoch_last_update) {
        const double step_size = 0.01;
        const double step = (add_rate_max - add_rate_min) / (curr_end - curr_start);
        if (curr_start <= epoch && epoch < curr_end) {
            *exact_rate = *exact_rate + step;
            *add_rate = step_size * py_float_floordiv(*exact_rate, step_size);
        }
        *epoch_last_update = epoch;
    }
}

/* TJ:206-252 step.  thr = floor(add_rate * 2^24) (oracle/philox.py rate_threshold).
 * Returns 0; -2 on bad action. reward float64. */
int orc_tj_step(const orc_tj_cfg* c, const int32_t* grid, const int32_t* route_off,
                const int32_t* route_rc, const int32_t* action, int32_t thr, uint32_t seed,
                uint32_t env_gid, uint32_t episode, uint32_t t,
                int32_t* alive, int32_t* wait, int32_t* loc, int32_t* last_act, int32_t* route_loc,
                int32_t* route_id, int32_t* cars_in_sys, int32_t* has_failed, int32_t* is_completed,
                double* reward)
{
    (void)grid;
    const int N = c->N;
    for (int i = 0; i < N; ++i) if (action[i] > 2) return -2;             /* TJ:228 (naction=2, <=) */
    memset(is_completed, 0, sizeof(int32_t) * (size_t)N);                 /* TJ:233 */
    /* TJ:540-581 _take_action */
    for (int i = 0; i < N; ++i) {
        if (alive[i] == 0) continue;                                      /* TJ:542-543 */
        wait[i] += 1;                                                     /* TJ:546 */
        if (action[i] == 1) { last_act[i] = 1; continue; }                /* TJ:549-551 */
        if (action[i] == 0) {                                             /* TJ:554 */
            route_loc[i] += 1;                                            /* TJ:556 */
            const int len = route_off[route_id[i] + 1] - route_off[route_id[i]];
            if (route_loc[i] == len) {                                    /* TJ:560-568 */
                *cars_in_sys -= 1; alive[i] = 0; wait[i] = 0;
                loc[2 * i] = loc[2 * i + 1] = 0; is_completed[i] = 1;
                continue;
            }
            const int32_t* cell = route_rc + 2 * (size_t)(route_off[route_id[i]] + route_loc[i]);
            loc[2 * i] = cell[0]; loc[2 * i + 1] = cell[1];               /* TJ:575-578 */
            last_act[i] = 0;                                              /* TJ:581 */
        }
    }
    /* TJ:369-393 _add_cars */
    for (int r = 0; r < c->narrival; ++r) {
        if (*cars_in_sys >= N) break;                                     /* TJ:371-372 (return) */
        const uint32_t u = orc_x24(seed, env_gid, DOMAIN_TJ_ADD, episode, t, 3u * r + 0);
        if ((int32_t)u <= thr) {                                          /* TJ:375 */
            int dead[1024], nd = 0;                                       /* TJ:614-618 _choose_dead */
            for (int i = 0; i < N; ++i) if (alive[i] == 0) dead[nd++] = i;
            const uint32_t ud = orc_x24(seed, env_gid, DOMAIN_TJ_ADD, episode, t, 3u * r + 1);
            const int idx = dead[(int)(((uint64_t)ud * (uint32_t)nd) >> 24)];
            alive[idx] = 1;                                               /* TJ:380 */
            const uint32_t up = orc_x24(seed, env_gid, DOMAIN_TJ_ADD, episode, t, 3u * r + 2);
            const int p_i = (int)(((uint64_t)up * (uint32_t)c->routes_per_arrival) >> 24);   /* TJ:383 */
            route_id[idx] = p_i + r * c->routes_per_arrival;              /* TJ:385 */
            route_loc[idx] = 0;                                           /* TJ:389 */
            const int32_t* cell = route_rc + 2 * (size_t)route_off[route_id[idx]];
            loc[2 * idx] = cell[0]; loc[2 * idx + 1] = cell[1];           /* TJ:390 */
            *cars_in_sys += 1;                                            /* TJ:393 */
        }
    }
    /* obs is taken here (TJ:240) by orc_tj_obs */
    /* TJ:585-595 _get_reward */
    for (int i = 0; i < N; ++i) {
        double r = -0.01 * (double)wait[i];                               /* TJ:586 */
        int same = 0;
        for (int j = 0; j < N; ++j)
            if (j != i && loc[2 * j] == loc[2 * i] && loc[2 * j + 1] == loc[2 * i + 1]) same = 1;
        if (same && (loc[2 * i] != 0 || loc[2 * i + 1] != 0)) {           /* TJ:589-592 l.any(), Q10 */
            r += -10.0;
            *has_failed = 1;
        }
        reward[i] = (double)alive[i] * r;                                 /* TJ:594 */
    }
    return 0;
}

/* TJ:321-366 _get_obs (vocab_type 'bool') + env_wrappers.py:88-100.
 * Row a: [last_act/(naction-1), route_id/(npath-1), window one-hot (W*W*vocab)], zero if dead.
 * CAR channel counts every car on the cell incl. dead ones parked at (0,0) (Q8). */
/* TJ:321-366 with vocab_type 'scalar': the one-hot base grid has 3 columns (outside, road, car), CAR += 1 per car
 * (TJ:326-327), the outside column is dropped (TJ:331-332), p_norm = p / (h-1, w-1) is inserted (TJ:344,361). */
static void tj_obs_scalar(const orc_tj_cfg* c, const int32_t* grid, const int32_t* alive, const int32_t* loc,
                          const int32_t* last_act, const int32_t* route_id, float* obs)
{
    const int N = c->N, v = c->vision, W = 2 * v + 1;
    const size_t od = 4 + (size_t)W * W * 2;
    memset(obs, 0, sizeof(float) * od * (size_t)N);
    for (int a = 0; a < N; ++a) {
        if (alive[a] == 0) continue;
        float* row = obs + od * (size_t)a;
        row[0] = (float)((double)last_act[a] / 1.0);
        row[1] = (float)((double)route_id[a] / (double)(c->npath - 1));
        row[2] = (float)((double)loc[2 * a] / (double)(c->h - 1));
        row[3] = (float)((double)loc[2 * a + 1] / (double)(c->w - 1));
        for (int dy = 0; dy < W; ++dy)
            for (int dx = 0; dx < W; ++dx) {
                const int gr = loc[2 * a] + dy - v, gc = loc[2 * a + 1] + dx - v;
                float* cell = row + 4 + (size_t)(dy * W + dx) * 2;
                const int inside = (gr >= 0 && gr < c->h && gc >= 0 && gc < c->w);
                cell[0] = (inside && grid[gr * c->w + gc] == 1) ? 1.0f : 0.0f;      /* ROAD_CLASS column */
                for (int p = 0; p < N; ++p)
                    if (loc[2 * p] == gr && loc[2 * p + 1] == gc) cell[1] += 1.0f;   /* CAR_CLASS column */
            }
    }
}

void orc_tj_obs(const orc_tj_cfg* c, const int32_t* grid, const int32_t* alive, const int32_t* loc,
                const int32_t* last_act, const int32_t* route_id, float* obs)
{
    if (c->scalar) { tj_obs_scalar(c, grid, alive, loc, last_act, route_id, obs); return; }
    const int N = c->N, v = c->vision, W = 2 * v + 1, vocab = c->vocab;
    const size_t od = 2 + (size_t)W * W * vocab;
    memset(obs, 0, sizeof(float) * od * (size_t)N);
    for (int a = 0; a < N; ++a) {
        if (alive[a] == 0) continue;                                      /* TJ:352-356 */
        float* row = obs + od * (size_t)a;
        row[0] = (float)((double)last_act[a] / (double)(2 - 1));          /* TJ:338 */
        row[1] = (float)((double)route_id[a] / (double)(c->npath - 1));   /* TJ:341 */
        for (int dy = 0; dy < W; ++dy)
            for (int dx = 0; dx < W; ++dx) {
                const int gr = loc[2 * a] + dy - v, gc = loc[2 * a + 1] + dx - v;
                float* cell = row + 2 + (size_t)(dy * W + dx) * vocab;
                const int id = (gr >= 0 && gr < c->h && gc >= 0 && gc < c->w) ? grid[gr * c->w + gc]
                                                                              : c->outside; /* TJ:317 */
                cell[id] += 1.0f;                                         /* TJ:319 */
                for (int p = 0; p < N; ++p)                               /* TJ:326-327 */
                    if (loc[2 * p] == gr && loc[2 * p + 1] == gc) cell[c->car_class] += 1.0f;
            }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Action sampling (action_utils.py:32-36): multinomial(exp(logp), 1) restated as inverse-CDF on  */
/* the injected uniform.  fp32, left-to-right cumulative sum, last action is the fallback.        */
/* ------------------------------------------------------------------------------------------- */
int32_t orc_sample_one(const float* logp, int A, uint32_t x24v)
{
    const float u = (float)x24v * (1.0f / 16777216.0f);
    float cdf = 0.0f;
    for (int a = 0; a < A - 1; ++a) {
        cdf += expf(logp[a]);
        if (u < cdf) return a;
    }
    return A - 1;
}

/* Batch form used by tests: logp [rows][A] with rows = E*N (row = e*N + n), draw = head*N + n. */
void orc_sample_actions(const float* logp, int A, int E, int N, int head, uint32_t seed,
                        uint32_t env_gid0, const int32_t* episode, const int32_t* t, int32_t* action,
                        float* chosen_logp)
{
    for (int e = 0; e < E; ++e)
        for (int n = 0; n < N; ++n) {
            const size_t row = (size_t)e * N + n;
            const uint32_t x = orc_x24(seed, env_gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)episode[e],
                                       (uint32_t)t[e], (uint32_t)(head * N + n));
            const int32_t a = orc_sample_one(logp + row * A, A, x);
            action[row] = a;
            if (chosen_logp) chosen_logp[row] = logp[row * A + a];
        }
}
