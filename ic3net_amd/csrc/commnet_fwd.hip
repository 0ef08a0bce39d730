// commnet_fwd.hip — the NON-recurrent CommNet module (comm.py:127-129, 179-205, 220-224, 228-239) after the encoder, all
// communication passes in ONE launch — ic3_commnet_forward:
//     x = tanh(enc)                    enc = encoder(obs) incl. its bias (caller: sparse gather or dense GEMM)
//     h_0 = x
//     h_{i+1} = tanh(x + f_modules[i](h_i) + C_modules[i](comm(h_i)))      i = 0 .. comm_passes - 1
//     out = [log_softmax(heads_k(h)) ... | value_head(h)]
// A workgroup owns 64 rows = whole envs (the communication block mixes the agents of one env only).  The tile [comm | h]
// lives in LDS as the A operand; pass i's [C_i | F_i] (H x 2H, packed by ic3_commnet_pack like the C weights of
// policy_step.hip) streams from L2 as the B operand of v_mfma_f32_32x32x2_f32 — exact fp32 — x and the accumulators stay
// in registers in the MFMA C/D layout, tanh on the hardware transcendental unit (|error| <= ~2e-7, bar 1e-5).
// This is f3 coverage ("other policy variants"), not the headline path: one resident workgroup per CU at
// H >= 128, no store pacing, plain loops.
//
// Round 4 — ic3_commnet_step: the WHOLE rollout iteration of trainer.py:43-108 for the non-recurrent module as one launch
// (KIND = IC3_ENV_PP / IC3_ENV_TJ instantiations of the same kernel): window descriptors of the tile's envs in LDS ->
// the dense observation rows of the state acted on (pp/tj_obs_store_run: every element evaluated and stored once) -> sparse
// encoder gather (comm.py:119) -> tanh -> the communication passes above -> heads, log_softmax -> Philox inverse-CDF draws
// (action_utils.py:32-36, the counters of ic3_env_sample_actions) -> env.step (pp/tj_step_lanes: the device functions the
// stand-alone step kernels and policy_step_kernel run).  Five launches per step before (encode, this kernel's policy part,
// draws, step, obs).
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "env_device.hpp"
#include "ic3_common.hpp"
#include "ps_common.hpp"

namespace ic3 {

typedef float cn_f32x4 __attribute__((ext_vector_type(4)));
typedef float cn_f32x16 __attribute__((ext_vector_type(16)));

struct CommnetArgs {
    const float* enc;          // [R][H]
    const float* wp;           // [passes][2H/8][H][2][4] packed [C_i | F_i]
    const float* bias;         // [passes][H]  C_i.bias + f_i.bias
    const float* head_w;       // [OT][H]
    const float* head_b;       // [OT]
    const int32_t* alive_in;   // [R] or null
    const int32_t* comm_in;    // [R] or null
    float* out;                // [R][OT]
    float* h_out;              // [R][H] or null: the final hidden state (tests; the tanh recurrence: h_t)
    const float* h_in;         // [R][H] or null.  Non-null = the tanh RECURRENCE of models.RNN (models.py:68-92, rnn_type 'MLP'):
                               // h_t = tanh(affine1(obs) + affine2(h_{t-1})): x = enc WITHOUT the tanh, h_0 = h_in (one pass, comm off)
    int E, N, EPT, passes, mode_avg, comm_zero, nheads, OT, a0, a1, a2, a3;
    int n_full, EPTs;          // tiles [0, n_full) hold EPT envs, the tiles behind them EPTs envs (<= 32 rows: one 32-row MFMA tile)
    // ic3_commnet_step (KIND != 0): the env side of the iteration
    const cn_f32x4* Wt;        // encoder.weight^T [obs_dim][H/4]
    const cn_f32x4* enc_bias;  // encoder.bias [H/4]
    const cn_f32x4* loc_table; // ic3_env_encode_table or null
    float* obs;                // [E][N][obs_dim] or null: rows of the state acted on
    int32_t* action;           // [nheads][R]
    int obs_dim, G, tile_words;
    const void* wp3;           // or null: per pass three exact bf16 planes of [C_i | F_i] in fragment order (ic3_commnet_pack_split)
    int auto_reset;            // env handle in auto-reset mode: an env with t == 0 starts an episode — nobody is dead yet and
                               // the gate is 0 (trainer.py:41-46, quirks Q21 / Q22); the module carries no other state
    uint32_t seed, gid0;
    const int32_t* episode;
    const int32_t* tstep;
    StepOut so;
    PPState pp;
    TJState tj;
};

// Wave priority (policy_step.hip: the phases around a matrix product are dependent chains on the tile's critical path, the
// product is throughput work): 3 outside the [comm | h] product, 0 inside.
#define CN_PRIO(p) __builtin_amdgcn_s_setprio(p)

// NARROW (round 6): one pass with the communication block off and split products — the IC baseline's stand-in and the tanh
// recurrence — keeps only the h half of the A tile (33.8 instead of 66.5 KB at H = 128; the heads' weights in a region of their
// own): THREE workgroups per CU instead of two, so that the obs stores of one tile overlap the matrix work / dependent chains of
// two others (the kernel has no in-stream store pacing: a tile's stores and its products do not overlap inside a workgroup).
template <int H, int KIND = 0, bool NARROW = false>
__global__ __launch_bounds__(2 * H, (H <= 128) ? (NARROW ? 3 : 2) : 1) void commnet_forward_kernel(const CommnetArgs a)
{
    CN_PRIO(3);
    constexpr int K = 2 * H, LDA = (NARROW ? H : K) + 4, LDA4 = LDA / 4, NT = 2 * H, H4 = H / 4, KB = K / 8, BM = 64;
    constexpr int HO = NARROW ? 0 : H, HO4 = HO / 4;             // column of the h half inside the tile
    IC3_DYNAMIC_LDS(float, smem);
    float* const As = smem;                                      // [BM][LDA]: cols [0,H) comm, [H,2H) h  (NARROW: h alone)
    cn_f32x4* const As4 = reinterpret_cast<cn_f32x4*>(smem);
    float* const sm = As + BM * LDA;                             // [BM] m_j = alive_j * comm_action_j
    float* const sscale = sm + BM;                               // [BM] per-env 1 / (n_alive - 1)
    int32_t* const sal = reinterpret_cast<int32_t*>(sscale + BM);   // [BM] alive flags
    float* const zl = reinterpret_cast<float*>(sal + BM);        // [BM][16] logits
    int32_t* const sact = reinterpret_cast<int32_t*>(zl + BM * 16);   // [BM] env action (head 0) of every row      (KIND != 0)
    uint32_t* const rmask = reinterpret_cast<uint32_t*>(sact + BM);   // [BM] window cells of a row that carry a count
    int32_t* const sep = reinterpret_cast<int32_t*>(rmask + BM); // [BM] episode counters of the tile's envs (Philox key)
    int32_t* const sts = sep + BM;                               // [BM] step counters
    int32_t* const tile = sts + BM;                              // env descriptors of the tile's envs
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int col = 32 * w + li, N = a.N;
    const int bid = blockIdx.x;
    const bool short_tile = bid >= a.n_full;                      // (plan_commnet_tiles: full tiles first, short ones behind)
    const int e0 = short_tile ? a.n_full * a.EPT + (bid - a.n_full) * a.EPTs : bid * a.EPT;
    const int nenv = min(short_tile ? a.EPTs : a.EPT, a.E - e0), rows = nenv * N;
    const bool two = rows > 32;                                  // the second 32-row MFMA tile holds rows (workgroup-uniform)
    const size_t r0 = (size_t)e0 * N;
    const int WW = (KIND == 0) ? 0 : (KIND == IC3_ENV_PP) ? (2 * a.pp.v + 1) * (2 * a.pp.v + 1) : (2 * a.tj.v + 1) * (2 * a.tj.v + 1);
    const int total = a.pp.Np + a.pp.nprey;
    const int nsegE = N * WW;
    const int tjw = tj_tile_words(N, WW);
    const float invN = 1.0f / (float)N;
    (void)total;
    (void)tjw;

    // ---- masks and per-env scale (comm.py:102-107,175-177,194-196) ---------------------------------------------------------
    if (tid < BM) {
        const bool in = tid < rows;
        int al = (in && a.alive_in) ? a.alive_in[r0 + tid] : 1;
        int cm = (in && a.comm_in) ? a.comm_in[r0 + tid] : 1;
        if constexpr (KIND != 0) {
            if (a.auto_reset && in && a.tstep[e0 + div_small(tid, invN)] == 0) {   // the env starts an episode at this step
                al = 1;
                cm = a.comm_in ? 0 : 1;                          // (a gated policy: gate 0; CommNet without a gate head talks)
            }
        }
        sm[tid] = in ? (float)(al * cm) : 0.f;
        sal[tid] = al;
    }
    if constexpr (KIND != 0) {
        // ---- window descriptors of the tile's envs (positions, then the per-cell table), Philox counters ----------------------
        if (tid < BM) {
            rmask[tid] = (WW <= 32) ? 0u : ~0u;
            if (tid < nenv) {
                sep[tid] = a.episode[e0 + tid];
                sts[tid] = a.tstep[e0 + tid];
            }
        }
        int2* ptab = reinterpret_cast<int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
        if constexpr (KIND == IC3_ENV_PP) {
            int32_t* psr = tile;
            int32_t* psc = tile + a.EPT * total;
            for (int i = tid; i < nenv * total; i += NT) {
                psr[i] = a.pp.loc_r[(size_t)e0 * total + i];
                psc[i] = a.pp.loc_c[(size_t)e0 * total + i];
            }
        } else {
            for (int i = tid; i < nenv * N; i += NT) {
                const int el = div_small(i, invN);
                tj_tile_load_car(tj_tile_at(tile + el * tjw, N), a.tj, e0 + el, i - el * N);
            }
        }
        __syncthreads();
        const float inv_nsegE = 1.0f / (float)max(nsegE, 1), inv_WW = 1.0f / (float)max(WW, 1);
        for (int sg = tid; sg < nenv * nsegE; sg += NT) {
            const int el = div_small(sg, inv_nsegE), q = sg - el * nsegE;
            int2 d;
            if constexpr (KIND == IC3_ENV_PP) {
                d = pp_tab_entry(tile + el * total, tile + a.EPT * total + el * total, q, a.pp.Np, total, a.pp.dim, a.pp.v);
                ptab[sg] = d;
            } else {
                const TJTile t = tj_tile_at(tile + el * tjw, N);
                d = tj_tab_entry(t, a.tj, q);
                t.tab[q] = d;
            }
            if (d.y != 0 && WW <= 32) {
                const int ag = div_small(q, inv_WW);
                atomicOr(&rmask[el * N + ag], 1u << (q - ag * WW));
            }
        }
        __syncthreads();
        // ---- the dense observation rows of the state this step acts on (trainer.py:49), every element stored once ------------
        // Round 6, TJ: the tile's contiguous chunk of the obs tensor (~96-98 % zeros) is ZERO-FILLED here with 16-byte stores — no
        // per-element evaluation (it was a quarter of this kernel's vector instructions, profiles/r06/pmc_commnet.txt) — and the few
        // non-zero entries are patched in behind the draws, after every wave has seen its zero stores complete (as
        // policy_step_kernel and tj_obs_fill_kernel do; bit-identical to ic3_env_observe).
        if constexpr (KIND == IC3_ENV_PP) {
            // (PP rows — 145 KB per env — keep the run-based store, every element evaluated and stored once: with the zero fill the
            //  wait for a tile's 872 KB of stores in front of the patches holds the workgroup's CU slot; IC launch 0.294 against 0.261 ms)
            if (a.obs) {
                const int vocab = a.pp.dim * a.pp.dim + 4;
                // the tanh recurrence streams h_in / h_out (84 MB per launch) next to the rows: with plain stores the rows wash them out of
                // the L2 (0.275 against 0.254 ms per PP-hard launch); without that traffic plain stores are the faster ones (IC: 0.245
                // against 0.259 ms) — same-box A/B, profiles/r06/commnet_ept_sweep.txt
                if ((vocab & 3) == 0 && a.h_in) pp_obs_store_run<true>(ptab, a.obs, e0, nenv, nsegE, vocab, tid, NT, 0, 1);
                else if ((vocab & 3) == 0) pp_obs_store_run<false>(ptab, a.obs, e0, nenv, nsegE, vocab, tid, NT, 0, 1);
                else pp_obs_store_run_scalar(ptab, a.obs, e0, nenv, nsegE, vocab, tid, NT, 0, 1);
            }
        } else if (a.obs) {
            const long long b0 = (long long)e0 * N * a.obs_dim;  // first float of the tile's chunk
            const long long L = (long long)nenv * N * a.obs_dim;
            float* out = a.obs + b0;
            const int head = (int)((4 - (b0 & 3)) & 3);
            const long long nb = (L - head) >> 2;
            const int tail = (int)((L - head) & 3);
            if (tid < head) out[tid] = 0.0f;
            if (tid < tail) out[head + 4 * nb + tid] = 0.0f;
            cn_f32x4* out4 = reinterpret_cast<cn_f32x4*>(out + head);
            const cn_f32x4 z4 = { 0.f, 0.f, 0.f, 0.f };
            const int o = (int)(((b0 + head) >> 2) & 63);        // 1 KiB-aligned wave stores
            long long j = tid - o;
            if (j < 0) j += NT;
            for (; j < nb; j += NT) out4[j] = z4;   // (plain stores: the rows are small — 21 KB per env — and the patches then meet their lines
                                                    //  in the L2; non-temporal: 0.146 against 0.143 ms per TJ-medium launch)
        }
        // ---- x = tanh(encoder(obs)) as a sparse gather (comm.py:119,127-129) -> h half ------------------------------------------
        const PtrRows encW = { a.Wt }, encL = { a.loc_table };
        for (int idx = tid; idx < BM * H4; idx += NT) {
            const int row = idx / H4, c4 = idx - row * H4;
            cn_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (row < rows) {
                const int el = div_small(row, invN), aa = row - el * N;
                if constexpr (KIND == IC3_ENV_PP) {
                    v = pp_encode_row_t(tile + el * total, tile + a.EPT * total + el * total, ptab + el * nsegE, aa, c4, H4, WW,
                                        a.pp.dim * a.pp.dim + 4, a.pp.dim, encW, a.enc_bias, encL, rmask[row]);
                } else {
                    v = tj_encode_row_t(tj_tile_at(tile + el * tjw, N), a.tj, aa, c4, H4, encW, a.enc_bias, encL, rmask[row]);
                }
                if (!a.h_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fast_tanh(v[q]);
                }
            }
            As4[row * LDA4 + HO4 + c4] = v;
        }
    } else {
    // ---- x = tanh(enc) -> h half (comm.py:127-129) ---------------------------------------------------------------------------
    for (int idx = tid; idx < BM * H4; idx += NT) {
        const int row = idx / H4, c4 = idx - row * H4;
        cn_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (row < rows) {
            v = *reinterpret_cast<const cn_f32x4*>(a.enc + (r0 + row) * H + 4 * c4);
            if (!a.h_in) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fast_tanh(v[q]);
            }
        }
        As4[row * LDA4 + HO4 + c4] = v;
    }
    }
    __syncthreads();
    for (int el = tid; el < nenv; el += NT) {
        int n_alive = 0;
        for (int j = 0; j < N; ++j) n_alive += sal[el * N + j];
        sscale[el] = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
    }
    float xr[2][16];                                             // x in the MFMA C/D layout, kept for every pass
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) xr[rt][reg] = As[(32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh) * LDA + HO + col];
    __syncthreads();
    if (a.h_in) {
        // the tanh recurrence: h_0 = the state that entered the step (zero for an env that starts an episode here: auto-reset)
        for (int idx = tid; idx < BM * H4; idx += NT) {
            const int row = idx / H4, c4 = idx - row * H4;
            cn_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (row < rows) {
                bool fresh = false;
                if constexpr (KIND != 0) fresh = a.auto_reset && a.tstep[e0 + div_small(row, invN)] == 0;
                if (!fresh) v = *reinterpret_cast<const cn_f32x4*>(a.h_in + (r0 + row) * H + 4 * c4);
            }
            As4[row * LDA4 + HO4 + c4] = v;
        }
        __syncthreads();
    }

    for (int pass = 0; pass < a.passes; ++pass) {
        // ---- comm_j = m_j (S_e - m_j h_j) scale_e (closed form of comm.py:181-205) -> comm half -----------------------------
        if constexpr (!NARROW) {
            const int c4 = tid % H4;
            for (int el = tid / H4; el < nenv; el += NT / H4) {
                const cn_f32x4* hp = As4 + (el * N) * LDA4 + H4 + c4;
                const float scl = sscale[el];
                cn_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
                if (!a.comm_zero)
                    for (int i = 0; i < N; ++i) S = mask_fma4(sm[el * N + i], hp[i * LDA4], S);   // (one component per instruction: ic3_common.hpp)
                for (int j = 0; j < N; ++j) {
                    const float m = a.comm_zero ? 0.f : sm[el * N + j];
                    As4[(el * N + j) * LDA4 + c4] = comm_out4(m, S, hp[j * LDA4], scl);
                }
            }
            for (int idx = rows * H4 + tid; idx < BM * H4; idx += NT) {
                const int row = idx / H4, c4p = idx - row * H4;
                As4[row * LDA4 + c4p] = cn_f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
        }
        __syncthreads();
        // ---- acc = [comm | h] . [C_i | F_i]^T : lane (li, lh) of wave w reads Wp[kb][32 w + li][lh] -> k = 8 kb + 4 lh + j --
        cn_f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][i] = 0.0f;
        if (NARROW || a.wp3) {
            // the product as nine exact bf16 x bf16 products per fp32 product (policy_step_kernel's gate_split):
            // the weights' three planes in fragment order, Wp3[pass][plane][kb16][wave][lane] x 16 bytes, one 16-k block ahead in
            // registers; the activations split per wave from the fp32 LDS tile (ps_split_frag).  9 x 32 cycles per 16 k against
            // 8 x 64 on the fp32 instruction.
            typedef __bf16 cn_bf16x8 __attribute__((ext_vector_type(8)));
            constexpr int KB16 = K / 16, NWv = H / 32;
            const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const ps_u32x4*>(a.wp3) + (size_t)pass * 3 * KB16 * NWv * 64,
                                                        (uint32_t)(3 * KB16 * NWv * 64 * 16));
            const int wu = __builtin_amdgcn_readfirstlane(w);
            auto wb = [&](int pl, int kb) __attribute__((always_inline)) {
                return __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, ((pl * KB16 + kb) * NWv + wu) * 1024, 0);
            };
            const int kb0 = (NARROW || a.comm_zero) ? KB16 / 2 : 0;   // (comm_mask_zero / the IC stand-in: the comm half is all zeros)
            constexpr int AOFF = NARROW ? H4 : 0;                // (NARROW: the tile starts at k = H)
            // the loop in two copies, chosen once per tile: a short tile (rows <= 32, plan_commnet_tiles) runs the first 32-row MFMA
            // tile only — a branch inside the loop cost the full tiles 5 % (profiles/r06/commnet_ept_sweep.txt)
            auto product = [&](auto two_c) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
                ps_u32x4 bq[3], bn[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[pl] = wb(pl, kb0);
                CN_PRIO(0);
#pragma unroll 2
                for (int kb = kb0; kb < KB16; ++kb) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bn[pl] = wb(pl, kb + 1);                  // (past the end: zeros, never used)
                    ps_u32x4 ap[2][3];
                    const cn_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh - AOFF;
                    const cn_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh - AOFF;
                    ps_split_frag(s0[0], s0[1], ap[0]);
                    if constexpr (TWO) ps_split_frag(s1[0], s1[1], ap[1]);
#pragma unroll
                    for (int pb = 0; pb < 3; ++pb)
#pragma unroll
                        for (int pa = 2; pa >= 0; --pa) {                                    // (least significant term first)
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cn_bf16x8, ap[0][pa]),
                                                                             __builtin_bit_cast(cn_bf16x8, bq[pb]), acc[0], 0, 0, 0);
                            if constexpr (TWO)
                                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cn_bf16x8, ap[1][pa]),
                                                                                 __builtin_bit_cast(cn_bf16x8, bq[pb]), acc[1], 0, 0, 0);
                        }
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[pl] = bn[pl];
                }
            };
            if (two) product(std::true_type{});
            else product(std::false_type{});
        } else if constexpr (!NARROW) {
        const cn_f32x4* wp = reinterpret_cast<const cn_f32x4*>(a.wp) + (size_t)pass * (K * H / 4) + col * 2 + lh;
        constexpr int CH = 8;
        static_assert(KB % CH == 0, "2H / 8 is a multiple of 8");
        cn_f32x4 cb[2][CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) cb[0][k] = wp[(size_t)k * H * 2];
        CN_PRIO(0);                                              // (the phases around the product run at priority 3)
#pragma unroll 1
        for (int ch = 0; ch < KB / CH; ch += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = ch + half;
                if (c + 1 < KB / CH) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) cb[(half + 1) & 1][k] = wp[(size_t)((c + 1) * CH + k) * H * 2];
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int kb = c * CH + k;
                    const cn_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
                    const cn_f32x4 a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cb[half][k][j], acc[0], 0, 0, 0);
                        if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], cb[half][k][j], acc[1], 0, 0, 0);
                    }
                }
            }
        }
        }
        CN_PRIO(3);
        __syncthreads();                                         // every wave has read the old h
        // ---- h' = tanh(x + F h + C comm + biases) -> h half (comm.py:222-224) ------------------------------------------------
        const float b = a.bias[pass * H + col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                As[lr * LDA + HO + col] = fast_tanh(xr[rt][reg] + acc[rt][reg] + b);
            }
        __syncthreads();
    }

    // ---- heads + value head (comm.py:228,239): weights -> rows [0, OT) of the comm half, logits -> zl ------------------------
    cn_f32x4* const hw4 = reinterpret_cast<cn_f32x4*>(tile + a.tile_words);          // NARROW: [16][H] the heads' weights
    for (int i = tid; i < a.OT * H4; i += NT) {
        if constexpr (NARROW) hw4[i] = reinterpret_cast<const cn_f32x4*>(a.head_w)[i];
        else As4[(i / H4) * LDA4 + i % H4] = reinterpret_cast<const cn_f32x4*>(a.head_w)[i];
    }
    if (a.h_out) {
        for (int idx = tid; idx < rows * H4; idx += NT) {
            const int row = idx / H4, c4 = idx - row * H4;
            *reinterpret_cast<cn_f32x4*>(a.h_out + (r0 + row) * H + 4 * c4) = As4[row * LDA4 + HO4 + c4];
        }
    }
    __syncthreads();
    for (int task = tid; task < rows * a.OT; task += NT) {
        const int row = task / a.OT, o = task - row * a.OT;
        const cn_f32x4* hp = As4 + row * LDA4 + HO4;
        const cn_f32x4* wo = NARROW ? hw4 + o * H4 : As4 + o * LDA4;
        cn_f32x4 s = { 0.f, 0.f, 0.f, 0.f };
        for (int k = 0; k < H4; ++k) s += hp[k] * wo[k];
        zl[row * 16 + o] = (s[0] + s[1]) + (s[2] + s[3]) + a.head_b[o];
    }
    __syncthreads();
    const int sizes[4] = { a.a0, a.a1, a.a2, a.a3 };
    for (int task = tid; task < rows * (a.nheads + 1); task += NT) {
        const int tr = task / (a.nheads + 1), hd = task - tr * (a.nheads + 1);
        const float* z = zl + tr * 16;
        float* orow = a.out + (r0 + tr) * a.OT;
        int off = 0;
        for (int i = 0; i < hd && i < a.nheads; ++i) off += sizes[i];
        if (hd == a.nheads) {
            orow[off] = z[off];
            continue;
        }
        const int A = sizes[hd];
        float mx = -INFINITY;
        for (int o = 0; o < A; ++o) mx = fmaxf(mx, z[off + o]);
        float sum = 0.0f;
        for (int o = 0; o < A; ++o) sum += __builtin_amdgcn_exp2f(1.4426950408889634f * (z[off + o] - mx));
        const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(sum);
        for (int o = 0; o < A; ++o) orow[off + o] = z[off + o] - lse;
        if constexpr (KIND != 0) {          // the draw of this (row, head): same counters and arithmetic as sample_actions_env_kernel
            const int el = div_small(tr, invN), n = tr - el * N;
            const uint32_t x = philox_x24(a.seed, a.gid0 + (uint32_t)(e0 + el), DOMAIN_SAMPLE, (uint32_t)sep[el],
                                          (uint32_t)sts[el], (uint32_t)(hd * N + n));
            const float u = (float)x * (1.0f / 16777216.0f);
            float cdf = 0.0f;
            int act = A - 1;
            for (int o = 0; o < A - 1; ++o) {
                cdf += expf(z[off + o] - lse);
                if (u < cdf) {
                    act = o;
                    break;
                }
            }
            a.action[(size_t)hd * ((size_t)a.E * N) + r0 + tr] = act;
            if (hd == 0) sact[tr] = act;
        }
    }
    if constexpr (KIND != 0) {
        // ---- the non-zero entries of the tile's obs rows, on top of the zero fill issued at the start ---------------------------
        if constexpr (KIND == IC3_ENV_TJ) {
            if (a.obs) {
                IC3_WAIT_VMEM();                                 // this wave's zero stores ...
                __syncthreads();                                 // ... and every other wave's have completed
                float* orow0 = a.obs + (size_t)e0 * N * a.obs_dim;
                const float inv_q = 1.0f / (float)(nsegE + N);
                for (int sg = tid; sg < nenv * (nsegE + N); sg += NT) {
                    const int el = div_small(sg, inv_q), q = sg - el * (nsegE + N);
                    tj_obs_patch(tj_tile_at(tile + el * tjw, N), a.tj, orow0 + (size_t)el * N * a.obs_dim, a.obs_dim, WW, q);
                }
            }
        }
        // ---- env.step for the tile's envs with the env-action head (env_wrappers.py:76-77) -------------------------------------
        __syncthreads();
        const int lgG = __builtin_ctz(a.G);
        for (int base = 0; base < a.EPT * a.G; base += NT) {
            const int lt = base + tid;
            const int el = lt >> lgG, n = lt - (el << lgG);
            const int e = el < nenv ? e0 + el : a.E;
            if constexpr (KIND == IC3_ENV_PP) {
                pp_step_lanes(a.pp, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
            } else {
                tj_step_lanes(a.tj, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
            }
        }
    }
}

// Wp[kb][col][hh][j] = W[col][8 kb + 4 hh + j], W = [C | F] (H x 2H): the layout the kernel's B fragments read
__global__ void commnet_pack_kernel(const float* __restrict__ Cw, const float* __restrict__ Fw, float* __restrict__ Wp, int H)
{
    const long long n = (long long)H * 2 * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), hh = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int colx = (int)(rest % H), kb = (int)(rest / H);
        const int k = 8 * kb + 4 * hh + j;
        Wp[i] = k < H ? Cw[(size_t)colx * H + k] : Fw[(size_t)colx * H + (k - H)];
    }
}

// gate_split's planes for [C | F] (H x 2H): Wp3[plane][kb16][wave][lane] = 8 x bf16 { W_plane[32 wave + li][16 kb16 + 8 lh + i] },
// the three planes an exact split of every weight (ps_split3)
__global__ void commnet_pack_split_kernel(const float* __restrict__ Cw, const float* __restrict__ Fw, ps_u32x4* __restrict__ Wp, int H)
{
    const int NWv = H / 32, KB16 = 2 * H / 16;
    const long long per = (long long)KB16 * NWv * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long rest = i >> 6;
        const int wv = (int)(rest % NWv), kb = (int)(rest / NWv);
        const int li = lane & 31, lh = lane >> 5;
        const size_t row = (size_t)32 * wv + li;
        unsigned p[3][8];
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * kb + 8 * lh + q;
            ps_split3(k < H ? Cw[row * H + k] : Fw[row * H + (k - H)], p[0][q], p[1][q], p[2][q]);
        }
        for (int pl = 0; pl < 3; ++pl) {
            ps_u32x4 v;
            for (int d = 0; d < 4; ++d) v[d] = p[pl][2 * d] | (p[pl][2 * d + 1] << 16);
            Wp[(size_t)pl * per + i] = v;
        }
    }
}

}  // namespace ic3

extern "C" int ic3_commnet_pack_split(const float* C_weight, const float* f_weight, void* wp3, int H, ic3_stream stream)
{
    using namespace ic3;
    if (!C_weight || !f_weight || !wp3 || H <= 0 || (H & 31)) return fail(-22, "ic3_commnet_pack_split: hid_size a positive multiple of 32");
    hipLaunchKernelGGL(commnet_pack_split_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, C_weight, f_weight,
                       reinterpret_cast<ps_u32x4*>(wp3), H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_commnet_forward_supported(int H, int N) { return (H == 64 || H == 128 || H == 256) && N >= 1 && N <= 64; }

extern "C" int ic3_commnet_pack(const float* C_weight, const float* f_weight, float* wp, int H, ic3_stream stream)
{
    using namespace ic3;
    if (!C_weight || !f_weight || !wp || H <= 0 || (H & 7)) return fail(-22, "ic3_commnet_pack: bad arguments");
    hipLaunchKernelGGL(commnet_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, C_weight, f_weight, wp, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

extern "C" int ic3_commnet_forward(const float* enc, int E, int N, int H, int comm_passes, const float* wp, const void* wp3,
                                   const float* bias,
                                   const float* head_w, const float* head_b, const int32_t* head_sizes, int nheads,
                                   int mode_avg, int comm_zero, const int32_t* alive_in, const int32_t* comm_in, float* out,
                                   float* h_out, ic3_stream stream)
{
    using namespace ic3;
    if (!enc || !wp || !bias || !head_w || !head_b || !head_sizes || !out || E <= 0 || comm_passes < 1)
        return fail(-22, "ic3_commnet_forward: bad arguments");
    if (!ic3_commnet_forward_supported(H, N))
        return fail(-38, "ic3_commnet_forward: needs hid_size 64/128/256 and <= 64 agents per env");
    if (nheads < 1 || nheads > 4) return fail(-22, "ic3_commnet_forward: 1..4 action heads");
    CommnetArgs a{};
    a.enc = enc;
    a.wp = wp;
    a.wp3 = wp3;
    a.bias = bias;
    a.head_w = head_w;
    a.head_b = head_b;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.h_out = h_out;
    a.E = E;
    a.N = N;
    a.EPT = 64 / N;
    a.passes = comm_passes;
    a.mode_avg = mode_avg;
    a.comm_zero = comm_zero;
    a.nheads = nheads;
    int sz[4] = { 0, 0, 0, 0 };
    a.OT = 1;
    for (int i = 0; i < nheads; ++i) {
        sz[i] = head_sizes[i];
        if (sz[i] < 1) return fail(-22, "ic3_commnet_forward: empty action head");
        a.OT += sz[i];
    }
    if (a.OT > 16) return fail(-22, "ic3_commnet_forward: more than 15 actions in total");
    a.a0 = sz[0];
    a.a1 = sz[1];
    a.a2 = sz[2];
    a.a3 = sz[3];
    const int tiles = (E + a.EPT - 1) / a.EPT;
    a.n_full = tiles;
    a.EPTs = a.EPT;
    const size_t lds = ((size_t)64 * (2 * H + 4) + 3 * 64 + 64 * 16 + 4 * 64) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define IC3_CN(h)                                                                                                 \
    case h:                                                                                                       \
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h, 0>), lds));               \
        hipLaunchKernelGGL((commnet_forward_kernel<h, 0>), dim3(tiles), dim3(2 * h), lds, s, a);                       \
        break;
    switch (H) {
        IC3_CN(64)
        IC3_CN(128)
        IC3_CN(256)
    }
#undef IC3_CN
    IC3_HIP(hipGetLastError());
    return 0;
}

// CUs of the current device (the tile plans below)
static int commnet_cus()
{
    static int cu_count[64] = { 0 };   // per device (a process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cu_count[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cu_count[dev] = prop.multiProcessorCount;
    }
    return cu_count[dev] ? cu_count[dev] : 256;
}

// Envs per tile of the NARROW Predator-Prey launch (the IC baseline's stand-in, the tanh recurrence).  That launch is bound by its
// obs stores (6 % of the matrix peak): a tile costs its bytes (~ its envs) plus a fixed part (descriptors, the weights' fragments,
// the dependent chain behind the product — measured ~0.15 of one env's stores), and the launch lasts as long as the CU with the most
// tiles: the dispatcher deals the workgroups round-robin, so a tile count that is no multiple of the CU count leaves some CUs one
// tile more than the others (8192 envs of 10 agents: 1366 tiles of 6 envs = 5.3 per CU against 2048 tiles of 4 envs = 8 per CU;
// 0.254-0.260 -> 0.241-0.247 ms per launch on the same box).  Half of that imbalance is charged: late tiles on emptier CUs run faster.  Results do not depend
// on the tile size (rows are independent; the draws are keyed per env).  profiles/r06/commnet_ept_sweep.txt: six env counts x
// five tile sizes against this choice.
static int plan_store_bound_ept(int E, int ept_max)
{
    const int cus = commnet_cus();
    int best = ept_max;
    double best_cost = 0.0;
    for (int ept = ept_max; ept >= 1 && ept >= ept_max / 3; --ept) {
        const int tiles = (E + ept - 1) / ept;
        const double per_cu = (double)tiles / cus, worst = (double)((tiles + cus - 1) / cus);
        const double cost = 0.5 * (per_cu + worst) * (ept + 0.15);
        if (ept == ept_max || cost < best_cost - 1e-9) {
            best = ept;
            best_cost = cost;
        }
    }
    return best;
}

// Tiles of the launches that are bound by their vector / matrix work (every instantiation but the narrow Predator-Prey one): a.EPT
// = 64 / N envs fill the 64-row tile, and a tile count that is no multiple of the CU count leaves some CUs a whole tile more than the
// others (8192 envs of 10 agents: 1366 tiles = 5.3 per CU; E = 7680 / 8192 / 9216 cost 0.137 / 0.152 / 0.156 ms on TJ-medium).  As
// policy_step_kernel's plan B: as many FULL tiles as give every CU the same number, the rest as SHORT tiles of 32 / N envs (one
// 32-row MFMA tile: half the matrix work and half the rows' vector work — measured 0.5 of a full tile) dispatched
// behind them; taken when at least one round of full tiles remains and the CU with the most work ends earlier.  Results do not
// depend on the plan (GPU test).  Returns the number of workgroups.
static int plan_commnet_tiles(ic3::CommnetArgs& a)
{
    const int n_all = (a.E + a.EPT - 1) / a.EPT;
    a.n_full = n_all;
    a.EPTs = a.EPT;
    const int cus = commnet_cus(), epts = 32 / a.N;
    if (epts < 1) return n_all;
    const int n_full = (a.E / a.EPT) / cus * cus;
    if (n_full < cus) return n_all;
    const int n_short = (a.E - n_full * a.EPT + epts - 1) / epts;
    if (n_full / cus + 0.5 * ((n_short + cus - 1) / cus) < (double)((n_all + cus - 1) / cus) - 1e-9) {
        a.n_full = n_full;
        a.EPTs = epts;
        return n_full + n_short;
    }
    return n_all;
}

static size_t commnet_step_tile_words(const ic3_env* env)
{
    const int N = env->dims.N, EPT = 64 / N, WW = env->dims.window * env->dims.window;
    size_t w;
    if (env->kind == IC3_ENV_PP) w = (size_t)((2 * EPT * (env->pp.N + env->pp.nprey) + 3) & ~3) + (size_t)2 * EPT * N * WW;
    else w = (size_t)EPT * (((7 * N + 3) & ~3) + 2 * N * WW);
    return (w + 3) & ~(size_t)3;
}

static size_t commnet_step_lds(const ic3_env* env, int H, bool narrow = false)
{
    return ((size_t)64 * ((narrow ? H : 2 * H) + 4) + 3 * 64 + 64 * 16 + 4 * 64 + commnet_step_tile_words(env) + (narrow ? 16 * H : 0)) *
           sizeof(float);
}

extern "C" int ic3_commnet_step_supported(const ic3_env* env, int H)
{
    if (!env || !ic3_commnet_forward_supported(H, env->dims.N)) return 0;
    const size_t lds = commnet_step_lds(env, H);
    return lds <= 160 * 1024 ? (int)lds : 0;
}

extern "C" int ic3_commnet_step(ic3_env* env, const float* enc_wt, const float* enc_bias, const float* loc_table, int H,
                                int comm_passes, const float* wp, const void* wp3, const float* bias, const float* head_w,
                                const float* head_b, const int32_t* head_sizes, int nheads, int mode_avg, int comm_zero, const int32_t* alive_in,
                                const int32_t* comm_in, const float* h_in, float* h_out, float* out, int32_t* action, float* obs,
                                float* reward, int32_t* done, int32_t* alive, int32_t* is_completed, ic3_stream stream)
{
    using namespace ic3;
    ic3::Range range_("ic3_commnet_step");
    if (h_in && (!h_out || h_out == h_in || comm_passes != 1 || !comm_zero))
        return fail(-22, "ic3_commnet_step: the tanh recurrence (h_in) takes one pass with the communication block off and h_out != h_in");
    if (!env || !enc_wt || !enc_bias || !wp || !bias || !head_w || !head_b || !head_sizes || !out || !action || !reward || !done ||
        comm_passes < 1)
        return fail(-22, "ic3_commnet_step: bad arguments");
    if (env->resets == 0) return fail(-22, "ic3_commnet_step: reset() has not been called");
    if (nheads < 1 || nheads > 4) return fail(-22, "ic3_commnet_step: 1..4 action heads");
    const int lds = ic3_commnet_step_supported(env, H);
    if (!lds)
        return fail(-38, "ic3_commnet_step: needs hid_size 64/128/256, <= 64 agents per env and an env tile that fits in LDS (use "
                         "ic3_env_encode + ic3_commnet_forward + ic3_env_sample_actions + ic3_env_step)");
    CommnetArgs a{};
    a.wp = wp;
    a.wp3 = wp3;
    a.bias = bias;
    a.head_w = head_w;
    a.head_b = head_b;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.h_in = h_in;
    a.h_out = h_out;
    a.E = env->dims.E;
    a.N = env->dims.N;
    a.EPT = 64 / a.N;
    a.passes = comm_passes;
    a.auto_reset = env->auto_max_steps > 0;
    a.mode_avg = mode_avg;
    a.comm_zero = comm_zero;
    a.nheads = nheads;
    int sz[4] = { 0, 0, 0, 0 };
    a.OT = 1;
    for (int i = 0; i < nheads; ++i) {
        sz[i] = head_sizes[i];
        if (sz[i] < 1) return fail(-22, "ic3_commnet_step: empty action head");
        a.OT += sz[i];
    }
    if (a.OT > 16) return fail(-22, "ic3_commnet_step: more than 15 actions in total");
    a.a0 = sz[0];
    a.a1 = sz[1];
    a.a2 = sz[2];
    a.a3 = sz[3];
    a.Wt = reinterpret_cast<const cn_f32x4*>(enc_wt);
    a.enc_bias = reinterpret_cast<const cn_f32x4*>(enc_bias);
    a.loc_table = reinterpret_cast<const cn_f32x4*>(loc_table);
    a.obs = obs;
    a.obs_dim = env->dims.obs_dim;
    a.action = action;
    a.tile_words = (int)commnet_step_tile_words(env);
    a.episode = env->f("episode");
    a.tstep = env->f("t");
    a.so = StepOut{ reward, done, alive, is_completed, env->d_err };
    const bool pp = env->kind == IC3_ENV_PP;
    if (pp) {
        a.pp = pp_state_of(env);
        a.G = group_lanes(a.N);
        a.seed = env->pp.seed;
        a.gid0 = env->pp.env_id_offset;
    } else {
        a.tj = tj_state_of(env);
        a.G = tj_group(a.N);
        a.seed = env->tj.seed;
        a.gid0 = env->tj.env_id_offset;
    }
    env->touch_obs(obs);
    hipStream_t s = (hipStream_t)stream;
    // one-shot (ic3_env_set_step_events): the dispatch itself stamps the caller's events
    hipEvent_t ev0 = (hipEvent_t)env->ev_start, ev1 = (hipEvent_t)env->ev_stop;
    env->ev_start = env->ev_stop = nullptr;
    // one pass, communication off, split products: the narrow tile (three workgroups per CU)
    const bool narrow = wp3 && comm_zero && comm_passes == 1;
    const size_t ldsn = commnet_step_lds(env, H, true);
    int tiles;
    if (narrow && pp && obs) {
        a.EPT = plan_store_bound_ept(a.E, a.EPT);
        tiles = (a.E + a.EPT - 1) / a.EPT;
        a.n_full = tiles;
        a.EPTs = a.EPT;
    } else {
        tiles = plan_commnet_tiles(a);
    }
#define IC3_CS(h)                                                                                                          \
    case h:                                                                                                                \
        if (narrow && pp) {                                                                                                \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h, IC3_ENV_PP, true>), ldsn)); \
            hipExtLaunchKernelGGL((commnet_forward_kernel<h, IC3_ENV_PP, true>), dim3(tiles), dim3(2 * h), ldsn, s, ev0, ev1, 0, a); \
        } else if (narrow) {                                                                                               \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h, IC3_ENV_TJ, true>), ldsn)); \
            hipExtLaunchKernelGGL((commnet_forward_kernel<h, IC3_ENV_TJ, true>), dim3(tiles), dim3(2 * h), ldsn, s, ev0, ev1, 0, a); \
        } else if (pp) {                                                                                                   \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h, IC3_ENV_PP>), lds));        \
            hipExtLaunchKernelGGL((commnet_forward_kernel<h, IC3_ENV_PP>), dim3(tiles), dim3(2 * h), lds, s, ev0, ev1, 0, a); \
        } else {                                                                                                           \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(commnet_forward_kernel<h, IC3_ENV_TJ>), lds));        \
            hipExtLaunchKernelGGL((commnet_forward_kernel<h, IC3_ENV_TJ>), dim3(tiles), dim3(2 * h), lds, s, ev0, ev1, 0, a); \
        }                                                                                                                  \
        break;
    switch (H) {
        IC3_CS(64)
        IC3_CS(128)
        IC3_CS(256)
    }
#undef IC3_CS
    IC3_HIP(hipGetLastError());
    return 0;
}
