/*
 * ic3_rollout.h — C ABI of libic3rollout.so, the MI355X-native batched rollout engine.
 *
 * The reference (IC3Net, /root/reference) is pure Python and has no FFI: its "plugin interface" for
 * this path is the duck-typed gym Env / GymWrapper / policy / Trainer surface.
 * This header is the boundary a maintainer binds *under* that surface (ctypes stub in
 * INTEGRATION.md; the build's own binding is ic3net_amd/_lib.py).  Each entry point cites the
 * reference interface it replaces.
 *
 * Conventions
 *   - return 0 = OK, negative errno-style code on error; ic3_last_error() gives the message.
 *   - every tensor argument is a DEVICE pointer owned by the caller (e.g. a torch tensor's
 *     data_ptr()); the library owns only the opaque handle (struct-of-arrays env state + constant
 *     tables).  No torch types cross this boundary.
 *   - all work is enqueued asynchronously on the caller's stream (hipStream_t passed as void*);
 *     no hidden synchronisation except where documented (get/set_state, stats: debug/parity).
 *   - a handle is bound to one device, one handle per process per GPU, not thread-safe.
 *   - layouts: per-agent arrays are [E][N] (row = e*N + n, env-major) int32/float32; per-env arrays
 *     are [E]; observations are [E][N][obs_dim] float32 — exactly the rows the policy's encoder
 *     GEMM consumes (replaces env_wrappers.py:88-100 `_flatten_obs`).
 *   - randomness: counter-based Philox4x32-10 keyed by (seed, env_id_offset + e); see DESIGN.md §RNG.
 */
#ifndef IC3_ROLLOUT_H
#define IC3_ROLLOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of THIS header.  The structs the caller fills (ic3_policy, ic3_episode) have grown between versions, so the
 * boundary checks itself: ic3_version() returns the library's value, ic3_abi_check() compares the caller's version and
 * struct sizes with the library's, and both structs start with `struct_size` (= sizeof, set by the caller) — an entry
 * point handed a struct of another size refuses it with -EINVAL before reading any other field. */
#define IC3_VERSION 601 /* 0.6.0 (round 6: ic3_comm_backward, ic3_lstm_weight_grad, ic3_bptt_backward; ic3_lstm_gates_backward_given takes the heads' share) */

/* 0 when `version` == IC3_VERSION of the library and the two sizes are the library's sizeof(ic3_policy) /
 * sizeof(ic3_episode); -EINVAL (with a message naming the mismatch) otherwise.  A binding calls it once after loading. */
int ic3_abi_check(int version, size_t sizeof_policy, size_t sizeof_episode);

typedef struct ic3_env ic3_env; /* opaque */
typedef void* ic3_stream;       /* hipStream_t */

enum { IC3_ENV_PP = 1, IC3_ENV_TJ = 2 };
enum { IC3_PP_MIXED = 0, IC3_PP_COOPERATIVE = 1, IC3_PP_COMPETITIVE = 2 }; /* --mode, predator_prey_env.py:67,261-269 */
enum { IC3_TJ_EASY = 0, IC3_TJ_MEDIUM = 1, IC3_TJ_HARD = 2 };             /* --difficulty, traffic_junction_env.py:74 */

/* Predator-Prey config == the args read by PredatorPreyEnv.multi_agent_init (predator_prey_env.py:72-110). */
typedef struct {
    int32_t E;             /* number of parallel environments in this handle (new: --nenvs) */
    int32_t N;             /* args.nfriendly (npredator)      :80 */
    int32_t nprey;         /* args.nenemies; only 1 supported (reference quirk Q7, :258) */
    int32_t dim;           /* args.dim                        :81 */
    int32_t vision;        /* args.vision                     :75 */
    int32_t mode;          /* IC3_PP_*                        :75,261-269 */
    int32_t stay;          /* !args.no_stay                   :82,90-93 */
    int32_t moving_prey;   /* must be 0: -ENOSYS otherwise (NotImplementedError :84-85) */
    int32_t enemy_comm;    /* args.enemy_comm :75: prey rows are appended to obs / reward (:203-207,255,276-281) and the
                              policy sees N + nprey agents (main.py:125-130); dims.N reports that row count */
    uint32_t seed;         /* Philox key[0] */
    uint32_t env_id_offset;/* global id of env 0 of this shard (multi-GPU sharding), Philox key[1] = offset+e */
} ic3_pp_cfg;

/* Traffic-Junction config == the args read by TrafficJunctionEnv.multi_agent_init (traffic_junction_env.py:80-158). */
typedef struct {
    int32_t E;
    int32_t N;             /* args.nagents (ncar)             :88 */
    int32_t dim;           /* args.dim                        :89 */
    int32_t vision;        /* args.vision                     :91 */
    int32_t difficulty;    /* IC3_TJ_*                        :90 */
    int32_t vocab_type;    /* 0 = 'bool', 1 = 'scalar' (:76,129-148): obs rows [last_act, route, r/(h-1), c/(w-1), (road, #cars) per cell] */
    double add_rate_min;   /* :103 */
    double add_rate_max;
    double curr_start;
    double curr_end;
    uint32_t seed;
    uint32_t env_id_offset;
} ic3_tj_cfg;

typedef struct {
    int32_t kind;          /* IC3_ENV_PP / IC3_ENV_TJ */
    int32_t E, N;
    int32_t obs_dim;       /* GymWrapper.observation_dim      env_wrappers.py:15-31 */
    int32_t vocab;         /* vocab_size                      PP:103 / TJ:134 */
    int32_t naction;       /* GymWrapper.num_actions          env_wrappers.py:33-40 */
    int32_t window;        /* 2*vision+1 */
    int32_t npath;         /* TJ:126 (0 for PP) */
    int32_t narrival;      /* len(routes) (0 for PP) */
    int32_t max_route_len; /* (0 for PP) */
    int32_t grid_h, grid_w;/* TJ self.dims (dim+1 for easy), PP dim */
    int32_t state_words;   /* int32 words of ic3_env_get_state/set_state */
} ic3_dims;

typedef struct {
    double success_sum;    /* PP: sum_e stat['success'] (PP:284-288); TJ: sum_e (1 - has_failed) (TJ:249) */
    double add_rate;       /* TJ stat['add_rate'] (TJ:250); 0 for PP */
    int64_t episodes;      /* number of resets so far * E */
    int64_t live_env_steps;/* env-steps actually simulated (not-yet-done envs) in the episodes now running */
    /* auto-reset mode (ic3_env_set_auto_reset): sums over the episodes that ENDED inside step launches since reset() */
    double auto_success_sum;  /* their stat['success'] */
    int64_t auto_episodes;    /* how many */
    int64_t auto_env_steps;   /* their lengths */
} ic3_stats;

int ic3_version(void);
const char* ic3_last_error(void); /* thread-local, valid until the next call on this thread */

/* gym.make('PredatorPrey-v0') + multi_agent_init(args): data.py:16-21, predator_prey_env.py:72-110 */
int ic3_pp_create(const ic3_pp_cfg* cfg, int device, ic3_env** out);
/* gym.make('TrafficJunction-v0') + multi_agent_init(args): data.py:22-27, traffic_junction_env.py:80-158;
 * builds grid ids and routes on the host (traffic_helper.py:5-209) and uploads them once. */
int ic3_tj_create(const ic3_tj_cfg* cfg, int device, ic3_env** out);
int ic3_env_destroy(ic3_env* env);
int ic3_env_dims(const ic3_env* env, ic3_dims* out);

/* Env.reset([epoch]) for all E envs: predator_prey_env.py:146-168 / traffic_junction_env.py:160-204.
 * epoch < 0 means "no epoch" (reset() without argument).  obs may be NULL. */
int ic3_env_reset(ic3_env* env, int epoch, float* obs, ic3_stream stream);

/* reset() into a given initial state (`init_state_or_null`): the reset bookkeeping (episode counters,
 * TJ curriculum, auto-reset accumulators) runs, then the integer state is replaced by `host_state` — a full dump in the
 * layout of ic3_env_get_state (`bytes` = dims.state_words * 4) — and obs, when given, is assembled from it.
 * host_state == NULL: plain ic3_env_reset.  Synchronises the stream (the host buffer is free on return). */
int ic3_env_reset_to(ic3_env* env, int epoch, const int32_t* host_state, size_t bytes, float* obs, ic3_stream stream);

/* Auto-reset: with max_steps > 0 an env whose episode ends at a step (episode_over, or max_steps steps played — the
 * trainer's forced done, trainer.py:90) starts its next episode inside the same step launch: episode += 1, t = 0, a
 * fresh state on the new episode's Philox key, exactly what a reset() would have drawn for that episode.  This is the
 * reference's collection loop (every finished episode is followed at once by the next one, trainer.py:107-108,
 * 227-242) without idling finished envs until a lock-step reset.  `done` then reports every episode end (1 also at the
 * max_steps cut); reward / alive / is_completed of that step belong to the finished episode, the next observation /
 * policy input to the new one; ic3_stats.auto_* accumulate the finished episodes' statistics.  ic3_policy_step in this
 * mode treats an env whose t == 0 as an episode start (h = c = 0, no alive mask, gate 0: trainer.py:38-46, quirks
 * Q21/Q22).  max_steps = 0 (default) restores lock-step episodes (finished envs freeze until ic3_env_reset). */
int ic3_env_set_auto_reset(ic3_env* env, int max_steps);
/* EXPERIMENT, off by default: incremental observation rows for ic3_policy_step.  With it on, the launch records per env
 * what it painted into the caller's obs buffer; the next launch that gets the SAME buffer — untouched since, which is the
 * caller's promise; writes through this handle (ic3_env_observe / step / reset with that buffer) are noticed — clears
 * exactly those entries and paints the new ones instead of zero-filling N*obs_dim floats per env.  The rows are
 * bit-identical either way; the HBM traffic is not (a few hundred partial-line writes instead of 145 KB per PP-hard env),
 * so bench lines measured with it are labelled and never the headline. */
int ic3_env_set_incremental_obs(ic3_env* env, int on);

/* Env.step(action) for all E envs: predator_prey_env.py:112-144 / traffic_junction_env.py:206-252.
 *   actions       [E][N] int32   the env-action head only (GymWrapper.step drops the talk head, env_wrappers.py:76-77)
 *   obs           [E][N][obs_dim] float32 or NULL
 *   reward        [E][N] float32 (the reference's float64 reward rounded to fp32)
 *   done          [E] int32      episode_over after this step
 *   alive         [E][N] int32 or NULL   info['alive_mask'] (TJ:244); ones for PP
 *   is_completed  [E][N] int32 or NULL   info['is_completed'] (TJ:247); zeros for PP
 * Environments whose episode is already over are frozen (the reference raises RuntimeError
 * "Episode is done", PP:129-130; a batched launch cannot raise per env): reward 0, done 1.
 * Out-of-range actions (> naction, the reference's assert PP:137 / TJ:228) set a sticky error flag
 * reported by ic3_env_check(). */
int ic3_env_step(ic3_env* env, const int32_t* actions, float* obs, float* reward, int32_t* done,
                 int32_t* alive, int32_t* is_completed, ic3_stream stream);

/* Observation of the current state without stepping (what reset/step return), obs [E][N][obs_dim]. */
int ic3_env_observe(ic3_env* env, float* obs, ic3_stream stream);
/* Same for the state held in a snapshot (ic3_env_snapshot; NULL = current state): lets the caller assemble the
 * observation of step t on a second stream while the policy / step kernels of step t+1 already run. */
int ic3_env_observe_at(ic3_env* env, const int32_t* snap, float* obs, ic3_stream stream);

/* encoder(obs(state)) without reading the observation back: out[e][n][:] = bias + sum_k obs[e][n][k] * Wt[k][:]
 * — the nn.Linear(obs_dim, hid) of comm.py:51,119 evaluated as a gather over the few non-zero obs entries
 * (<= 3 per window cell for PP, 2 + 2 per cell for TJ).  Wt = encoder.weight transposed, [obs_dim][H] row-major,
 * bias [H], out [E][N][H]; H % 4 == 0.  Mathematically identical to obs @ Wt + bias (fp32 sum order differs). */
int ic3_env_encode(ic3_env* env, const float* Wt, const float* bias, const float* loc_table /* or NULL */, float* out,
                   int ldo /* out row stride in floats, 0 = H */, int H, ic3_stream stream);
/* Optional accelerator for ic3_env_encode: the one-hot location channels of all window cells depend only on the
 * agent's grid position, so their W*W gathered rows can be pre-summed per position once per weight version:
 * loc_table [grid_h*grid_w][H] (dims.grid_h/w; PP: dim x dim), loc_table[pos] = sum_cells Wt[col(cell, id(pos, cell))].
 * With a table the encode gathers 1 + (#occupied cells) rows per agent instead of W*W + (#occupied cells). */
int ic3_env_encode_table(ic3_env* env, const float* Wt, int H, float* loc_table, ic3_stream stream);
/* ic3_env_encode for the state held in `snap` (ic3_env_snapshot; NULL = the live state): the update half re-evaluates
 * the encoder of step t from the state snapshot of that step instead of keeping its output (or the 145 KB observation
 * per env) from the rollout — replaces the saved input of comm.py:119's nn.Linear under autograd. */
int ic3_env_encode_at(ic3_env* env, const int32_t* snap, const float* Wt, const float* bias, const float* loc_table /* or NULL */,
                      float* out, int ldo, int H, ic3_stream stream);

/* Backward of ic3_env_encode for the update half (trainer.py:128-225 backpropagates through comm.py:51,119's
 * nn.Linear): given grad_out = dL/d out [E][N][H] (row stride ldg floats, 0 = H) it overwrites
 *   dWt   [obs_dim][H] = obs^T x grad_out   (the gradient of encoder.weight, transposed)
 *   dbias [H]          = sum of grad_out rows (may be NULL)
 * for the observation of the state held in `snap` (a device copy of the integer state taken with ic3_env_snapshot
 * at the time of the forward; NULL = the current state).  `work` is caller-owned scratch of
 * ic3_env_encode_backward_work(env, H) floats.  Mathematically identical to the dense product (fp32 atomics: the
 * summation order, hence the last ulp, varies from run to run). */
int ic3_env_snapshot(const ic3_env* env, int32_t* snap /* device, dims.state_words int32 */, ic3_stream stream);
int64_t ic3_env_encode_backward_work(const ic3_env* env, int H);
int ic3_env_encode_backward(ic3_env* env, const int32_t* snap, const float* grad_out, int ldg, int H, float* dWt,
                            float* dbias, float* work, ic3_stream stream);
/* The same gradient over SEVERAL states (the steps of an episode) with the expansion done once: the operation is linear in
 * what its first stage collects (per grid position / per shared column the sum of the grad_out rows standing there), so a
 * backward pass through time calls _accumulate once per step — `first` != 0 on the first call: the partial sums in `work`
 * are written, otherwise added to — and _finish once at the end, which overwrites dWt / dbias like ic3_env_encode_backward
 * would for the sum over all the accumulated states.  Same `work` (ic3_env_encode_backward_work floats) in every call of
 * the sequence.  -38 when the configuration's first stage does not run in its partial-sums form (grids too large for LDS):
 * the caller then uses ic3_env_encode_backward per step. */
int ic3_env_encode_backward_accumulate(ic3_env* env, const int32_t* snap, const float* grad_out, int ldg, int H, float* work,
                                       int first, ic3_stream stream);
int ic3_env_encode_backward_finish(ic3_env* env, int H, float* dWt, float* dbias /* or NULL */, float* work, ic3_stream stream);

/* The first stage over a WINDOW of T recorded states in ONE launch, on the matrix cores (enc_bwd.hpp, third form): the state of
 * step t at snaps + t * snap_words, its grad_out rows at grad_out + t * step_stride + row * ldg floats (ic3_bptt.dxh_step keeps
 * the per-step input gradients of a backward pass for this).  The position sums are one-hot x grad_out products — the one-hot
 * entries made in registers, grad_out split exactly into three bf16 terms, fp32 accumulation — and the shared columns' weights a
 * per-batch LDS table, split the same way.  `first` != 0 writes the partials in `work`
 * (ic3_env_encode_backward_window_work(env, H) floats; 0 = this configuration has no window form: hid_size a multiple of 32, and
 * of 128 above 128), otherwise adds to them; _window_finish expands them into dWt / dbias like _finish does for the per-step form
 * (the two forms' partials have different shapes: finish with the one that accumulated).  Reproducible run to run (no atomics in
 * stage 1; the expand stage's fp32 atomics as above). */
int64_t ic3_env_encode_backward_window_work(const ic3_env* env, int H);
int ic3_env_encode_backward_window(ic3_env* env, const int32_t* snaps, int64_t snap_words, int T, const float* grad_out, int ldg,
                                   int64_t step_stride, int H, float* work, int first, ic3_stream stream);
int ic3_env_encode_backward_window_finish(ic3_env* env, int H, float* dWt, float* dbias /* or NULL */, float* work, ic3_stream stream);

/* Synchronising: returns -EINVAL if any step since the last check saw an out-of-range action. */
int ic3_env_check(ic3_env* env, ic3_stream stream);

/* Parity / debug (synchronising, HOST pointers): full integer state.  Field layout by name:
 *   PP: "loc_r" "loc_c" [E][N+nprey], "reached" [E][N], "over" "success" "episode" "t" [E]
 *   TJ: "alive" "wait" "loc_r" "loc_c" "last_act" "route_loc" "route_id" "is_completed" [E][N],
 *       "cars_in_sys" "has_failed" "over" "episode" "t" [E]
 * ic3_env_state_field gives (offset, count) in int32 words inside the dump. */
int ic3_env_get_state(const ic3_env* env, int32_t* host_out, size_t bytes, ic3_stream stream);
int ic3_env_set_state(ic3_env* env, const int32_t* host_in, size_t bytes, ic3_stream stream);
int ic3_env_state_field(const ic3_env* env, const char* name, int64_t* offset_words, int64_t* count_words);

/* TJ constant tables (host copies, for parity with traffic_helper.get_routes): grid [h*w] road ids,
 * route_off [npath+1], route_rc [2*route_off[npath]].  Pass NULL to query sizes via ic3_env_dims. */
int ic3_tj_get_tables(const ic3_env* env, int32_t* grid, int32_t* route_off, int32_t* route_rc, size_t rc_capacity_words);

/* Host-only (no GPU needed): build the TJ tables for (dim, vision, difficulty) — what ic3_tj_create uploads.
 * Fills dims_out (grid_h/w, vocab, obs_dim, npath, narrival, max_route_len; E/N left 0) and, when non-NULL,
 * grid / route_off / route_rc.  Returns the number of int32 words route_rc needs, or a negative error
 * (-EINVAL with the reference's assert text for invalid dims, traffic_junction_env.py:93-100). */
int ic3_tj_build_tables(int dim, int vision, int difficulty, ic3_dims* dims_out, int32_t* grid, int32_t* route_off,
                        int32_t* route_rc, size_t rc_capacity_words);

/* TJ curriculum scalar state (traffic_junction_env.py:103-104,196-200,620-626) */
int ic3_tj_get_add_rate(const ic3_env* env, double* add_rate, double* exact_rate);

/* Measurement support (no reference counterpart): HIP events stamped by the ic3_policy_step dispatch itself.
 * ic3_env_set_step_events arms the NEXT ic3_policy_step launch on this handle (one shot): `start` is stamped when the
 * kernel begins, `stop` when it ends — no record packets in the stream around the launch.  Read the pair with
 * ic3_event_elapsed_ms after the stream has been synchronised. */
int ic3_event_create(void** event);
int ic3_event_destroy(void* event);
int ic3_event_elapsed_ms(void* start, void* stop, float* ms);
int ic3_env_set_step_events(ic3_env* env, void* start, void* stop);

/* Reduced episode statistics (synchronising; host struct): env.stat (PP:284-288, TJ:249-250). */
int ic3_env_stats(ic3_env* env, ic3_stats* host_out, ic3_stream stream);

/* What get_episode derives per step next to the policy / env calls (trainer.py:70-105,109-110), for n lock-step
 * slots of E envs, as one launch over the step-major episode buffers the step launches wrote:
 *   live[t][e]            1 while env e is still running when slot t starts (auto_reset: always 1)
 *   alive_mask[t][e][j]   alive * live                          (trainer.py:78-81)
 *   episode_mask[t][e]    0 where the transition ends the episode (done, or the last of max_steps slots: forced_last,
 *                         trainer.py:90-96), else 1
 *   episode_mini_mask     1 - is_completed unless the transition is done (trainer.py:98-99, quirk Q26)
 *   live_after[e]         live[n-1] * (1 - done[n-1])
 *   stats (device)        [0] num_steps = sum live, [1] envs with done set in slot n-1, [2, 2+N) per-agent reward sums
 *                         (trainer.py:86), [2+N, 2+2N) per-agent comm-action sums over live slots (trainer.py:73-75;
 *                         gate == NULL and !gate_ones: zeros), fp64, summed in a fixed order.
 * alive / is_completed may be NULL (PP: all ones / not reported).  scratch: ic3_episode_scratch_bytes(E, N) bytes;
 * counter: reserved (may be NULL).  Two launches (derivations + block partials, then a one-block fixed-order
 * reduction); asynchronous on `stream`. */
typedef struct ic3_episode {
    uint32_t struct_size;         /* sizeof(ic3_episode) of the caller's header (checked: -EINVAL on mismatch) */
    int32_t n, E, N;
    int32_t auto_reset, forced_last, gate_ones;
    const int32_t* done;          /* [n][E] */
    const int32_t* alive;         /* [n][E][N] or NULL */
    const int32_t* is_completed;  /* [n][E][N] or NULL */
    const float* reward;          /* [n][E][N] */
    const int32_t* gate;          /* talk-head actions, slot t at gate + t * gate_stride, [E][N] each; or NULL */
    int64_t gate_stride;
    float* live;                  /* [n][E] */
    float* alive_mask;            /* [n][E][N] */
    float* episode_mask;          /* [n][E] */
    float* episode_mini_mask;     /* [n][E][N] */
    float* live_after;            /* [E] */
    double* stats;                /* [2 + 2N] */
    double* scratch;
    int32_t* counter;
} ic3_episode;
size_t ic3_episode_scratch_bytes(int E, int N);
int ic3_episode_finalize(const ic3_episode* ep, ic3_stream stream);
/* The reversed return scan of compute_grad — replaces the loop trainer.py:162-171 and the mix of trainer.py:171 — over T
 * slots (the batch's transitions in time order, [T][E][N] step-major like the episode buffers) in one launch:
 *   coop[t] = reward[t] + gamma * coop[t+1] * episode_mask[t];   ncoop[t] = reward[t] + gamma * ncoop[t+1] * episode_mask[t] *
 *   episode_mini_mask[t];   returns[t][e][n] = mean_ratio * mean_n coop[t][e][:] + (1 - mean_ratio) * ncoop[t][e][n]
 * reward, episode_mini_mask, returns [T][E][N] f32; episode_mask [T][E] f32.  N <= 256. */
/* compute_grad's losses and the gradients they hand back to the policy's outputs (trainer.py:173-218) in ONE launch (round 6):
 * out [T][E*N][OT] = the rows the step launches wrote ([log-probs of every head | value]), action [T][nheads][E*N], returns
 * [T][E*N] (ic3_returns_scan's), alive_mask [T][E*N] (already times live), live [T][E]; advantages = (returns - value - adv_shift) *
 * adv_scale (normalize_rewards: the caller's mean and 1 / std over the live entries; else 0 and 1).  d_out [T][E*N][OT] receives
 * dL/d[logits of every head | value] with the log-softmax folded in (gradients w.r.t. its input), L = action_loss + value_coeff *
 * value_loss - entr * entropy; sums [ic3_loss_gradients_partials(T, E*N)][3] doubles = per-workgroup partial sums of (action_loss,
 * value_loss, entropy): the caller adds them up. */
int ic3_loss_gradients_partials(long long T, long long R);
int ic3_loss_gradients(const float* out, const int32_t* action, const float* returns, const float* alive_mask, const float* live,
                       const int32_t* head_sizes, int nheads, float adv_shift, float adv_scale, float entr, float value_coeff,
                       float* d_out, double* sums, int T, int E, int N, ic3_stream stream);
int ic3_returns_scan(const float* reward, const float* episode_mask, const float* episode_mini_mask, float gamma, float mean_ratio,
                     float* returns, int T, int E, int N, ic3_stream stream);

/* CommNetMLP communication block, comm.py:181-205, in closed form per env (SURVEY B.5 i):
 *   m_j = alive_j * comm_action_j ; out_j = m_j * (sum_i m_i h_i - m_j h_j) [ / (n_alive - 1) if mode_avg && n_alive > 1 ]
 * n_alive = sum_j alive_j (NOT the talker count, quirk Q23) or N when alive == NULL (quirk Q21).
 *   h [E][N][H] f32, alive [E][N] int32 or NULL, comm_action [E][N] int32 or NULL (all talk),
 *   out [E][N][H] f32.  mask_self: 1 = ones-eye comm_mask (default), 0 = comm_mask_zero (out = 0). */
int ic3_comm_masked_mean(const float* h, int ldh /* h row stride in floats, 0 = H */, const int32_t* alive,
                         const int32_t* comm_action, float* out, int E, int N, int H, int mode_avg, int mask_self,
                         ic3_stream stream);
/* The same block with an addend: out = addend + comm(h) — the backward of the communication block is the block itself applied
 * to the gradient (its mixing matrix is symmetric), and what it is added to (dL/dh of the recurrent path) comes along in the
 * same pass.  addend [E*N][H] rows with stride lda floats (0 = H; may be a column slice of a wider buffer).  out_row_scale
 * [E*N] or NULL: every output row times its factor (collection mode, trainer.py:227-242 through :128-225: the gradient that
 * would cross an episode boundary is dropped where it is produced).  H, ldh, lda multiples of 4. */
int ic3_comm_masked_mean_add(const float* h, int ldh, const int32_t* alive, const int32_t* comm_action, const float* addend,
                             int lda, const float* out_row_scale /* or NULL */, float* out, int E, int N, int H, int mode_avg,
                             int mask_self, ic3_stream stream);

/* Pointwise half of torch.nn.LSTMCell (comm.py:61,215; gate order i,f,g,o): gates [R][4H] already hold
 * W_ih x + b_ih + W_hh h + b_hh (two fp32 MFMA GEMMs, or one over [x | h]).  c [R][H] is updated in place,
 * h' is written to h_out with row stride ldh.  H % 4 == 0. */
int ic3_lstm_cell(const float* gates, float* c, float* h_out, int ldh, int R, int H, ic3_stream stream);
/* Its backward for the update half (trainer.py:128-225 backpropagating through comm.py:215): from the RE-COMPUTED gate
 * pre-activations gates [R][4H], the cell state c_prev [R][H] that entered the step, dh = dL/dh' [R][H] and dc = dL/dc'
 * [R][H] (NULL = zeros) -> dgates [R][4H] = dL/dgates (gate order i,f,g,o) and dc_prev [R][H] = dL/dc_prev (may alias
 * dc).  dbias (or NULL): [IC3_LSTM_BWD_MAX_PARTIALS][4H] scratch; the call WRITES its first n rows with partial column
 * sums of dgates and returns n > 0 — their sum is dL/db_ih = dL/db_hh.  Nothing of the forward has to be kept besides
 * (h, c) of every step.  H/4 a power of two <= 64.  Returns n (or 1 without dbias), negative errno on error. */
#define IC3_LSTM_BWD_MAX_PARTIALS 2048
int ic3_lstm_cell_backward(const float* gates, const float* c_prev, const float* dh, const float* dc /* or NULL */,
                           float* dgates, float* dc_prev, float* dbias /* or NULL */, int R, int H, ic3_stream stream);
/* The same gradient WITHOUT the pre-activations in memory (hid_size 64 / 128 / 256: ic3_lstm_gates_backward_supported):
 * one launch re-computes gates = [inp | h_prev] . [W_ih | W_hh]^T + bias on the fp32 matrix cores — xh [R][ldx] holds the
 * row [inp (H) | h_prev (H)] of comm.py:215's LSTMCell call (h_prev != NULL: [R][H], the h half is read from there and
 * written into xh by the same launch), lstm_wp is ic3_policy_pack's packed weight, bias [4H] = b_ih + b_hh — and applies the cell's derivative in the epilogue: dgates [R][4H], dc_prev [R][H] (may alias dc) as above.
 * dbias_partials (or NULL): [ceil(R / 64)][4H]; row w receives the column sums of dgates over rows [64 w, 64 w + 64) —
 * written when accumulate == 0, ADDED to what the row holds when accumulate != 0 (a whole episode's bias gradient then
 * needs one reduction at its end).  Returns the number of partial rows, negative errno on error (-38: unsupported H). */
int ic3_lstm_gates_backward_supported(int H);
int ic3_lstm_gates_backward(float* xh, int ldx, const float* h_prev /* or NULL */, const float* lstm_wp,
                            const void* lstm_wp3 /* NULL (fp32 instruction), or ic3_policy_pack_split's planes (exact bf16 split products) */,
                            const float* bias,
                            const float* c_prev,
                            const float* dh, const float* dc /* or NULL */, float* dgates, float* dc_prev,
                            float* dbias_partials /* or NULL */, int accumulate, int R, int H, ic3_stream stream);
/* The same launch + the INPUT gradient of the gate product in it (round 5): dxh [R][2H] = [d inp | d h_prev] = dgates .
 * [W_ih | W_hh] for the tile the workgroup holds anyway — replaces the (R x 4H) x (4H x 2H) product that followed the call and
 * its re-read of dgates.  Same arithmetic as the split gate product (nine exact bf16 x bf16 products per fp32 product, fp32
 * accumulation); lstm_wp3_bwd = ic3_policy_pack_split_bwd's planes (3 * 4H * 2H * 2 bytes).  Needs lstm_wp3 (split mode),
 * hid_size 64 / 128. */
int ic3_policy_pack_split_bwd(const float* w_ih /* [4H][H] */, const float* w_hh /* [4H][H] */, void* lstm_wp3_bwd, int H,
                              ic3_stream stream);
int ic3_lstm_gates_backward_dx(float* xh, int ldx, const float* h_prev /* or NULL */, const float* lstm_wp, const void* lstm_wp3,
                               const void* lstm_wp3_bwd, const float* bias, const float* c_prev, const float* dh,
                               const float* dc /* or NULL */, float* dgates, float* dc_prev, float* dbias_partials /* or NULL */,
                               int accumulate, float* dxh, int R, int H, ic3_stream stream);
/* The cell's derivative from RECORDED gates (round 5): `gates` [R][4H] = the activated i | f | g | o of the step as the
 * rollout's launch stored them (ic3_env_set_gates_out) — the gate product is not run again (it was the largest item of the
 * update half: 2 * R * 2H * 4H flop x 9 split products per recorded step).  Same outputs as ic3_lstm_gates_backward_dx; xh /
 * h_prev (both or neither): the launch copies h_prev into the h half of xh [R][ldx] for the weight-gradient product that
 * follows; lstm_wp3_bwd / dxh (both or neither): the input gradient in the same launch.  Collection mode (trainer.py:227-242:
 * a record slot where some envs start an episode, or after which the recurrent gradient must not pass): row_live [R] or NULL —
 * c_prev and the copied h_prev of a row are multiplied by it (0 = the env starts an episode at this slot: zero state);
 * row_keep [R] or NULL — dc of a row is multiplied by it (0 = nothing arrives from the next slot).  hid_size 64 / 128.
 * Round 6: dgates may BE gates (every lane overwrites exactly what it read: the record turns into the weight-gradient
 * product's operand in place); dhead [R][OT] / w_heads [OT][H] (both or neither, OT <= 16): the heads' share of dL/dh_t —
 * dhead . w_heads with dhead = dL/d[logits of every head | value] (comm.py:228,239), w_heads = heads.k.weight stacked, then
 * value_head.weight — is added to dh on the way in, instead of by an R x OT x H product (and a pass over dh) in front; dh and / or
 * dc may be NULL = zeros (a detach point of the recurrence, trainer.py:56-60: nothing arrives from the next step — no memset, no
 * read). */
int ic3_lstm_gates_backward_given(const float* gates, float* xh /* or NULL */, int ldx, const float* h_prev /* or NULL */,
                                  const void* lstm_wp3_bwd /* or NULL */, const float* c_prev, const float* dh,
                                  const float* dc /* or NULL */, float* dgates, float* dc_prev, float* dbias_partials /* or NULL */,
                                  int accumulate, float* dxh /* or NULL */, const float* row_live /* or NULL */,
                                  const float* row_keep /* or NULL */, const float* dhead /* or NULL */,
                                  const float* w_heads /* or NULL */, int OT, int R, int H, ic3_stream stream);

/* The communication block's and C's share of the backward through one recorded step (trainer.py:128-225 through
 * comm.py:181-206), one launch.  With M the per-env mixing matrix of ic3_comm_masked_mean (symmetric), comm = M h_prev and
 * inp = encoder(obs) + comm . C.weight^T:
 *     dh_out [E*N][H] = (d h_direct + (M d inp) . C.weight) * out_scale        — dL/dh_{t-1}, what step t - 1 receives
 *     C.weight's gradient += (M d inp)^T . h_prev                               (= d inp^T . comm: the forward's comm is not formed again)
 * dxh [E*N][ldd]: columns [0, H) = d inp, [H, 2H) = d h_direct (ic3_lstm_gates_backward_given's dxh); alive / gate [E][N] int32 or
 * NULL as ic3_comm_masked_mean; c_weight [H][H] = C_modules[0].weight as stored; out_scale [E*N] or NULL (collection mode: the
 * gradient that must not cross an episode boundary or a detach point is dropped where it is produced); dcw_partials
 * [ic3_comm_backward_partials(E, N)][H][H]: one partial per workgroup, written (accumulate == 0) or added to — their sum over dim 0
 * is the gradient.  comm_zero != 0 (comm.py:40-41: C sees zeros): dh_out = d h_direct * out_scale, nothing else is read.
 * Exact fp32 products on the fp32 matrix instruction.  hid_size 64 / 128, <= 64 agents per env.  Returns the number of partials
 * written (0 with comm_zero), negative errno on error. */
int ic3_comm_backward_partials(int E, int N);
int ic3_comm_backward(const float* dxh, int ldd, const float* h_prev, const int32_t* alive /* or NULL */,
                      const int32_t* gate /* or NULL */, const float* c_weight, const float* out_scale /* or NULL */, float* dh_out,
                      float* dcw_partials, int accumulate, int E, int N, int H, int mode_avg, int comm_zero, ic3_stream stream);

/* The LSTM cell's weight gradient over a whole WINDOW of recorded steps in one launch (+ a fixed-order reduction): dW [2H][4H]
 * (+)= [inp | h_prev]^T . dgates over Q = steps x rows pairs — dW = the gradient of [weight_ih | weight_hh]^T of comm.py:61's
 * LSTMCell (rows [0, H): weight_ih^T, rows [H, 2H): weight_hh^T).  inp [Q][ldi] (the first H floats of a row: the inp half of the
 * rollout's record, ic3_env_set_record_out), h_prev [Q][H] (slots 0..T-1 of the recorded hidden states — no copy into an
 * [inp | h] buffer), dgates [Q][4H] (what ic3_lstm_gates_backward_given left in the gate record), row_live [Q] or NULL (collection
 * mode: h_prev rows times it).  scratch: ic3_lstm_weight_grad_scratch_floats(Q, H) floats.  split != 0 (what ic3net_amd passes
 * with args.gate_split, the default): every fp32 operand split exactly into three bf16 terms, all nine cross products on the bf16
 * matrix cores, fp32 accumulation — the arithmetic of the rollout's gate product (ic3_policy.gate_split); split == 0: the fp32 matrix
 * instruction.  Exact products either way, split-K over the CUs, slices summed in order (reproducible).  hid_size 64 / 128. */
size_t ic3_lstm_weight_grad_scratch_floats(long long Q, int H);
int ic3_lstm_weight_grad(const float* inp, int ldi, const float* h_prev, const float* dgates, const float* row_live /* or NULL */,
                         long long Q, int H, float* dW, int accumulate, int split, float* scratch, ic3_stream stream);

/* The backward through a window of T recorded steps (trainer.py:128-225 over comm.py:134-244, one communication pass, recorded
 * gates), last step first, as ONE host call — per step: ic3_lstm_gates_backward_given (in place on the gate record, the heads'
 * share folded in, the input gradient in the same launch) -> ic3_comm_backward -> ic3_env_encode_backward_accumulate on the step's
 * snapshot; nothing runs on the host between the launches.  Afterwards the caller runs ic3_lstm_weight_grad over the window
 * (gates now holds dgates), ic3_heads_grad, ic3_env_encode_backward_finish and sums the partials.
 *   gates [T][R][4H] in: the recorded activated gates, out: dgates;  hs, cs [>= T][R][H] the state ENTERING every step;
 *   dhead [T][R][OT];  snaps: T snapshots, snap_words int32 apart;  alive / gate: HOST arrays of T device pointers ([E][N] int32,
 *   entries may be NULL) or NULL;  row_live / row_keep [T][R] or NULL (collection mode, as ic3_lstm_gates_backward_given; the
 *   communication backward of step t scales its output by row_keep[t - 1]);  detach_gap > 0: dh, dc are zeroed in front of every
 *   step t with (t + 1) % detach_gap == 0 (trainer.py:56-60, lock-step windows);  dh, dc [R][H] in: dL/d(h, c) arriving at the
 *   window's last step, out: leaving its first;  dxh [R][2H] scratch (or a ring of T of them: dxh_step);  dbias_partials [ceil(R / 64)][4H] and dcw_partials
 *   [ic3_comm_backward_partials][H][H] are ADDED to (zero them before the first window);  enc_work as
 *   ic3_env_encode_backward_accumulate, enc_first != 0: this window starts the accumulation;  gate_events: see the struct.
 * ic3_bptt_backward_supported(env, H): 1 when every step can run (hid_size 64 / 128, <= 64 agents, the encoder backward in its
 * partial-sums form) — the loop overwrites the record as it goes, so ask first. */
typedef struct ic3_bptt {
    uint32_t struct_size;   /* sizeof(ic3_bptt) of the caller's header (checked: -EINVAL on mismatch) */
    int32_t T, E, N, H, OT;
    int32_t mode_avg, comm_zero, detach_gap, enc_first;
    float* gates;
    const float* hs;
    const float* cs;
    const float* dhead;
    const int32_t* snaps;
    int64_t snap_words;
    const int32_t* const* alive;
    const int32_t* const* gate;
    const float* row_live;
    const float* row_keep;
    const void* lstm_wp3_bwd;
    const float* w_heads;
    const float* c_weight;
    float* dh;
    float* dc;
    float* dxh;
    float* dbias_partials;
    float* dcw_partials;
    float* enc_work;
    int64_t dxh_step;       /* 0: dxh is one [R][2H] buffer and every step runs ic3_env_encode_backward_accumulate on it;  > 0: dxh is
                               a ring, step t's input gradients at dxh + t * dxh_step floats (>= R * 2H), and the encoder's first
                               stage runs ONCE behind the loop over all T of them (ic3_env_encode_backward_window: enc_work of
                               ic3_env_encode_backward_window_work floats, finish with ic3_env_encode_backward_window_finish) */
    int32_t two_chains;     /* != 0 (with dxh_step): the steps of envs [0, E1) and [E1, E), E1 = ic3_bptt_first_chain_envs(E, N), are
                               launched on two streams — the caller's and one the library owns, forked / joined with events — so that
                               one chain's workgroups fill the ragged last round of the other's launches.  dcw_partials then holds
                               ic3_comm_backward_partials(E1, N) + ic3_comm_backward_partials(E - E1, N) slots */
    void** gate_events;     /* measurement support: NULL, or 2 T events (ic3_event_create) — [2t] / [2t + 1] are recorded on the
                               stream in front of / behind step t's gate launch (read them with ic3_event_elapsed_ms) */
} ic3_bptt;
int ic3_bptt_backward_supported(const ic3_env* env, int H);
int ic3_bptt_first_chain_envs(int E, int N);
int ic3_bptt_backward(ic3_env* env, const ic3_bptt* b, ic3_stream stream);
/* The weight / bias gradient of the heads + value head over a whole episode in one pass (trainer.py:128-225 through
 * comm.py:228,239): dW [OT][H] += sum_m d[m][o] h[m][c], db [OT] += sum_m d[m][o] over the M = steps x rows pairs
 * (d [M][OT], h [M][H]: h_t of every step, i.e. the recorded hidden states shifted by one step).  scratch:
 * ic3_heads_grad_scratch_floats(H) floats.  Fixed-order reduction (reproducible).  hid_size 64 / 128 / 256. */
size_t ic3_heads_grad_scratch_floats(int H);
int ic3_heads_grad(const float* d, const float* h, long long M, int H, int OT, float* dW, float* db, float* scratch,
                   ic3_stream stream);

/* Action heads + value head + log_softmax (comm.py:228,239) in one pass: out[r][:] =
 * [log_softmax(W_0 h_r + b_0) | ... | log_softmax(W_{k-1} h_r + b_{k-1}) | w_v h_r + b_v], OT = sum A_k + 1 <= 16.
 * W [OT][H] = rows of heads.k.weight stacked, then value_head.weight; b [OT] likewise; head_sizes is a HOST array. */
int ic3_policy_heads(const float* h, int ldh, const float* W, const float* b, const int32_t* head_sizes, int nheads,
                     float* out, int R, int H, ic3_stream stream);

/* ic3_lstm_cell + ic3_policy_heads (+ ic3_env_sample_actions for every head when `action` is non-NULL) in one launch:
 * the lanes that produce a row of h' keep it in registers for the head / value dot products.  out [R][sum A_k + 1] as
 * ic3_policy_heads; action [nheads][R] int32 (action_utils.py:32-36 draws, Philox counters (episode, t) read from
 * `env`, which must satisfy E*N == R) or NULL.  Returns -ENOSYS unless H/4 is a power of two <= 64. */
int ic3_lstm_cell_heads(const float* gates, float* c, float* h_out, int ldh, int R, int H, const float* W, const float* b,
                        const int32_t* head_sizes, int nheads, float* out, const ic3_env* env, int32_t* action,
                        ic3_stream stream);

/* select_action (action_utils.py:32-36): one multinomial draw per (env, agent) row from exp(logp),
 * as inverse-CDF on Philox uniforms: counter (head*N+n, t, episode, DOMAIN_SAMPLE), key (seed, env_id_offset+e).
 *   logp [E*N rows][ld] f32 (first A columns of each row) -> action [E][N] int32, chosen_logp [E][N] f32 or NULL. */
int ic3_sample_actions(const float* logp, int ld /* logp row stride in floats, 0 = A */, int A, int head, uint32_t seed,
                       uint32_t env_id_offset, uint32_t episode, uint32_t t, int32_t* action, float* chosen_logp,
                       int E, int N, ic3_stream stream);

/* The same draw with (seed, env_id_offset) taken from the handle's config and (episode, t) read from its
 * device-side per-env counters (the env's own episode / step counters): every argument is constant across
 * steps, so a hipGraph capture of the rollout step can be replayed for the whole run. */
int ic3_env_sample_actions(const ic3_env* env, const float* logp, int ld, int A, int head, int32_t* action,
                           float* chosen_logp, ic3_stream stream);

/* ---------------------------------------------------------------------------------------------------------------
 * One launch per rollout step: the body of the reference's hot loop trainer.py:43-108 for all E envs —
 *     action_out, value, (h, c) = policy_net([state, (h, c)], info)      comm.py:134-244 (recurrent CommNet / IC3Net,
 *                                                                          one communication pass)
 *     action = select_action(args, action_out)                            action_utils.py:32-36 (Philox draws as
 *                                                                          ic3_env_sample_actions, every head)
 *     next_state, reward, done, info = env.step(action[0])                PP:112-144 / TJ:206-252
 * The policy reads the observation through the env's integer state (sparse encoder, ic3_env_encode), never through
 * a dense obs tensor.  A workgroup owns 64/N whole envs; encoder output, communication vectors, gate pre-activations
 * and logits stay in LDS / registers (fp32 MFMA, exact f32 products).
 *
 * The ic3_policy struct holds DEVICE pointers, all float32:
 *   enc_wt    [obs_dim][H]  encoder.weight^T                       comm.py:51
 *   enc_bias  [H]           encoder.bias + C_modules[0].bias        comm.py:119,206 (quirk Q24)
 *   loc_table [grid][H]     ic3_env_encode_table(enc_wt) or NULL
 *   c_wp      [H*H]         C_modules[0].weight, packed by ic3_policy_pack
 *   lstm_wp   [4H*2H]       [f_module.weight_ih | weight_hh], packed by ic3_policy_pack
 *   lstm_bias [4H]          bias_ih + bias_hh
 *   head_w    [OT][H]       heads.k.weight stacked, then value_head.weight; head_b [OT] likewise; OT = sum A_k + 1 <= 16
 *   mode_avg  args.comm_mode == 'avg' (comm.py:194);  comm_zero  args.comm_mask_zero (comm.py:40-41)
 * Arguments:
 *   h, c       [E*N][H]  LSTM state, updated in place                 trainer.py:49-60
 *   alive_in   [E][N] int32 or NULL   info['alive_mask'] of the PREVIOUS step (NULL at t = 0, quirk Q21)
 *   comm_in    [E][N] int32 or NULL   info['comm_action'] (gate sampled at t-1, zeros at t = 0, quirk Q22; NULL = all talk)
 *   out        [E*N][OT]  log_softmax of every head, then the value
 *   action     [nheads][E][N] int32   the draws of every head
 *   obs        [E][N][obs_dim] or NULL   the dense observation of the state this call ACTS ON — the `state` the reference
 *              hands to policy_net at this step (trainer.py:49), i.e. what ic3_env_observe would return BEFORE the call —
 *              written by the same launch: the tile's slice is zero-filled by non-temporal stores issued between the
 *              MFMAs and its few non-zero entries are patched in at the end (bit-identical to ic3_env_observe).  The
 *              observation of the NEW state is what the next call writes (or ic3_env_observe).  When the descriptors do
 *              not fit in LDS the library launches ic3_env_observe in front of the kernel instead: same contents.
 *   reward / done / alive / is_completed   as ic3_env_step
 * In auto-reset mode (ic3_env_set_auto_reset) an env whose t == 0 is at an episode start: its h, c count as zero, its
 * alive mask as absent and its gate as 0, whatever the arguments hold (trainer.py:38-51).
 * Returns -ENOSYS when ic3_policy_step_supported(env, H) == 0 (H not in {64,128,256}, > 64 agents, or an env tile
 * that does not fit in LDS): use the separate entry points then. */
typedef struct {
    uint32_t struct_size;   /* sizeof(ic3_policy) of the caller's header (checked: -EINVAL on mismatch) */
    int32_t H;
    int32_t nheads;
    int32_t head_sizes[4];
    int32_t mode_avg;
    int32_t comm_zero;
    const float* enc_wt;
    const float* enc_bias;
    const float* loc_table;
    const float* c_wp;
    const float* lstm_wp;
    const float* lstm_bias;
    const float* head_w;
    const float* head_b;
    /* comm_passes > 1 (comm.py:179-218: the communication block + C_i + f_module run `comm_passes` times per step, each
     * pass on the hidden state the previous one left).  One call = one pass: pass_index = i, c_wp / enc_bias = those of
     * C_modules[i] (enc_bias = encoder.bias + C_i.bias); inner_pass != 0 for every pass but the last — such a call only
     * updates h, c (out / action / obs / reward / done / alive / is_completed may be NULL, the env does not step).  The
     * last pass is the ordinary call.  pass_index > 0 also tells an auto-reset handle that h, c are this step's, not
     * the previous episode's.  Both 0: the one-pass policy of the BASELINE configs. */
    int32_t pass_index;
    int32_t inner_pass;
    /* gate_split != 0 with lstm_wp3 = ic3_policy_pack_split's buffer: the gate product [inp | h] . [W_ih | W_hh]^T with every
     * fp32 operand split EXACTLY into three bf16 terms and all nine cross products on the bf16 matrix cores (each product
     * exact in fp32, fp32 accumulation: fp32-class arithmetic — measured error against fp64 = the fp32 matrix instruction's,
     * DESIGN.md section 0; what ic3net_amd passes by default).  0: the fp32 matrix instruction.  Honoured by ic3_policy_step
     * and ic3_policy_forward. */
    int32_t gate_split;
    /* npasses >= 2 (round 5): EVERY communication pass of the step inside ONE ic3_policy_step launch — pass i uses
     * c_wp_pass[i] / enc_bias_pass[i] (those of C_modules[i], as above); h stays in LDS between the passes (only the last
     * pass's h' goes to memory), c passes through c_out.  pass_index and inner_pass must be 0; needs gate_split and
     * hid_size 64 / 128, at most 4 passes (-ENOSYS otherwise: one call per pass as above).  0 or 1: one pass per call. */
    int32_t npasses;
    const void* lstm_wp3;
    const float* c_wp_pass[4];
    const float* enc_bias_pass[4];
} ic3_policy;

/* The NON-recurrent CommNet module after the encoder (comm.py:127-129,179-205,220-224,228-239), every communication pass
 * in ONE launch:  x = tanh(enc);  h_0 = x;  h_{i+1} = tanh(x + f_modules[i](h_i) + C_modules[i](comm(h_i)));
 * out [E*N][OT] = [log_softmax(heads_k(h)) ... | value_head(h)].  enc [E*N][H] = encoder(obs) including its bias
 * (ic3_env_encode or a dense GEMM); wp = comm_passes blocks of 2*H*H floats, block i = ic3_commnet_pack(C_modules[i].weight,
 * f_modules[i].weight); bias [comm_passes][H] = C_i.bias + f_i.bias; head_w / head_b / head_sizes as ic3_policy_heads;
 * alive_in / comm_in as ic3_policy_step (NULL = everyone alive / talking); h_out [E*N][H] or NULL receives the final
 * hidden state.  hid_size 64 / 128 / 256, <= 64 agents per env (ic3_commnet_forward_supported), else -ENOSYS.
 * wp3 != NULL (comm_passes blocks of 3 * 2*H*H bf16 = 3*H*H floats, block i = ic3_commnet_pack_split of the same weights; what
 * ic3net_amd passes with args.gate_split, the default): the product [comm | h] . [C_i | F_i]^T with every fp32 operand split exactly
 * into three bf16 terms, all nine cross products on the bf16 matrix cores, fp32 accumulation — the arithmetic of
 * ic3_policy.gate_split; NULL: the fp32 matrix instruction on wp. */
int ic3_commnet_forward_supported(int H, int N);
int ic3_commnet_pack(const float* C_weight /* [H][H] */, const float* f_weight /* [H][H] */, float* wp /* [2*H*H] */, int H,
                     ic3_stream stream);
int ic3_commnet_pack_split(const float* C_weight /* [H][H] */, const float* f_weight /* [H][H] */, void* wp3 /* 3 * 2*H*H * 2 bytes */,
                           int H, ic3_stream stream);
int ic3_commnet_forward(const float* enc, int E, int N, int H, int comm_passes, const float* wp, const void* wp3 /* or NULL */,
                        const float* bias,
                        const float* head_w, const float* head_b, const int32_t* head_sizes, int nheads, int mode_avg,
                        int comm_zero, const int32_t* alive_in, const int32_t* comm_in, float* out, float* h_out /* or NULL */,
                        ic3_stream stream);

/* The whole rollout iteration of trainer.py:43-108 for the NON-recurrent module as ONE launch — what ic3_policy_step is for
 * the recurrent policy: the sparse encoder on the env's integer state (enc_wt [obs_dim][H] = encoder.weight^T, enc_bias [H] =
 * encoder.bias, loc_table = ic3_env_encode_table(enc_wt) or NULL), ic3_commnet_forward's passes / heads / log_softmax (same
 * wp / wp3 / bias / head_* arguments), the action draws of every head (action [nheads][E][N], Philox counters of
 * ic3_env_sample_actions), env.step with head 0 (reward / done / alive / is_completed as ic3_env_step) and, when obs != NULL,
 * the dense observation rows [E][N][obs_dim] of the state this call ACTS ON (bit-identical to ic3_env_observe before the
 * call).  -ENOSYS when ic3_commnet_step_supported(env, H) == 0 (hid_size not 64/128/256, > 64 agents, a tile that does
 * not fit in LDS).  Handles in auto-reset mode are taken (an env that starts an episode: nobody dead, gate 0).
 * h_in != NULL (round 6): the tanh RECURRENCE of the IRIC baseline (models.py:68-92, rnn_type 'MLP') as the same launch —
 * h_t = tanh(affine1(obs) + affine2(h_{t-1})): enc_wt / enc_bias = affine1, block 0 of wp / wp3 = pack(zeros, affine2.weight), bias =
 * affine2.bias, one pass, comm_zero != 0; h_in [E*N][H] the state entering the step (rows of an env that starts an episode are read
 * as zero in auto-reset mode), h_out [E*N][H] (!= h_in) receives h_t.  h_in == NULL: h_out, when given, receives the module's final
 * hidden state. */
int ic3_commnet_step_supported(const ic3_env* env, int H);
int ic3_commnet_step(ic3_env* env, const float* enc_wt, const float* enc_bias, const float* loc_table /* or NULL */, int H,
                     int comm_passes, const float* wp, const void* wp3 /* or NULL */, const float* bias, const float* head_w,
                     const float* head_b, const int32_t* head_sizes, int nheads, int mode_avg, int comm_zero, const int32_t* alive_in,
                     const int32_t* comm_in, const float* h_in /* or NULL */, float* h_out /* or NULL */, float* out, int32_t* action,
                     float* obs /* or NULL */, float* reward, int32_t* done, int32_t* alive, int32_t* is_completed, ic3_stream stream);

int ic3_policy_pack(const float* C_weight /* [H][H] */, const float* w_ih /* [4H][H] */, const float* w_hh /* [4H][H] */,
                    float* c_wp /* H*H */, float* lstm_wp /* 4H*2H */, int H, ic3_stream stream);
int ic3_policy_pack_split(const float* w_ih /* [4H][H] */, const float* w_hh /* [4H][H] */, void* lstm_wp3 /* 3 * 2H * 4H * 2 bytes */,
                          int H, ic3_stream stream);
/* Diagnostic: the gate product of the LSTM cell ALONE — gates [R][4H] = xh [R][2H] . [W_ih | W_hh]^T, no bias — through the
 * operand layouts, activation split and per-accumulator instruction order of ic3_policy_step's gate loops: lstm_wp3 == NULL:
 * the fp32 matrix instruction on ic3_policy_pack's lstm_wp; lstm_wp3 != NULL: the exact bf16 split products on
 * ic3_policy_pack_split's planes.  For measuring what the two arithmetic modes do on operands no rollout produces. */
int ic3_gate_product_probe(const float* xh, const float* lstm_wp, const void* lstm_wp3, float* gates, int R, int H,
                           ic3_stream stream);
/* One-shot: the NEXT ic3_policy_step on this handle (its final communication pass) reads the LSTM state from its h / c
 * arguments as always but writes the new state to h_out / c_out [E*N][H] instead of updating h / c in place — a caller that
 * keeps the state ENTERING every step of an episode (the update half, trainer.py:128-225: backward through time over
 * recorded (h, c)) lets the launch write slot t + 1 of its record directly instead of copying 2 x E*N*H floats per step. */
int ic3_env_set_hidden_out(ic3_env* env, float* h_out, float* c_out);
/* One-shot as well: the NEXT ic3_policy_step also stores what the update half's backward would otherwise compute again
 * (trainer.py:128-225 over a recorded rollout): `gates` [E*N][4H] = the activated gates of its LSTM cell — sigmoid(i) |
 * sigmoid(f) | tanh(g) | sigmoid(o), exactly the values its cell update used — for ic3_lstm_gates_backward_given; and, when
 * `xh` is not NULL, the inp half of the rows of xh [E*N][2H] (row stride 2H; inp = encoder(obs) + C(comm) + both biases, the
 * left operand of the gate product — the h half is not touched: ic3_lstm_gates_backward_given copies h_prev there).
 * Needs ic3_policy.gate_split, one communication pass per launch, hid_size 64 / 128 (-38 from the step call otherwise);
 * gates = NULL disarms.  Costs 16 H (+ 4 H) bytes of stores per agent row in the launch; nothing when not armed. */
int ic3_env_set_record_out(ic3_env* env, float* gates, float* xh /* or NULL */);
int ic3_policy_step_supported(const ic3_env* env, int H); /* 0, or the LDS bytes per workgroup */
/* The policy half alone, for callers that bring their own encoder output (a dense observation that is not an env's
 * current state, comm.py:119 evaluated as a GEMM): enc [E*N][H] = encoder(x) + C.bias -> out [E*N][OT] as above, h / c
 * updated in place; p->enc_wt / enc_bias / loc_table are not read.  No draws, no env step.  H in {64,128,256}, N <= 64. */
int ic3_policy_forward(const ic3_policy* p, const float* enc, int E, int N, float* h, float* c, const int32_t* alive_in,
                       const int32_t* comm_in, float* out, ic3_stream stream);
int ic3_policy_step(ic3_env* env, const ic3_policy* p, float* h, float* c, const int32_t* alive_in,
                    const int32_t* comm_in, float* out, int32_t* action, float* obs, float* reward, int32_t* done,
                    int32_t* alive, int32_t* is_completed, ic3_stream stream);

/* Synthetic uniform actions in [0, naction) for env-only benchmarks (DOMAIN_BENCH). */
int ic3_random_actions(int32_t* action, int naction, uint32_t seed, uint32_t env_id_offset, uint32_t episode,
                       uint32_t t, int E, int N, ic3_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* IC3_ROLLOUT_H */
