// policy_step.hip — ONE launch per rollout step (gfx950): the whole iteration of /root/reference/trainer.py:43-108
// for a tile of whole environments,
//     CommNetMLP.forward (comm.py:134-244: encoder -> masked communication mean -> C -> LSTMCell -> heads, log_softmax)
//     select_action      (action_utils.py:32-36, Philox inverse-CDF)
//     env.step           (predator_prey_env.py:112-144 / traffic_junction_env.py:206-252, without the observation)
// Communication only mixes agents of the SAME env (comm.py:181-205), so a workgroup that owns EPT = 64/N whole envs
// (<= 64 agent rows) needs nothing from any other workgroup: the encoder output, the communication vectors, `inp`,
// the (rows x 4H) gate pre-activations and the logits only ever exist in LDS / registers.  HBM traffic per row:
// read h, c (2H floats) + write h', c' (2H) + log-probs/value/actions/reward — the six-kernel chain it replaces moved
// ~11H floats per row through HBM (enc, comm, inp, gates written and re-read).
//
// MFMA-bound: 2*R*(2H*4H + H*H) flops on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate; 157 TFLOP/s
// peak).  Decomposition for H = 128 (256 threads, 2 workgroups per CU: what they overlap is each other's non-matrix
// phases — an fp32 MFMA stream leaves the other waves of its SIMD no issue slots):
//   * tile = 64 rows x ALL 4H gate columns; wave w owns hidden columns [32w, 32w+32) of the four gates -> 2 (row
//     tiles) x 4 (gates) accumulators of 32x32; a lane holds the SAME (row, column) of all four gates, so the LSTM
//     nonlinearity needs no cross-lane traffic;
//   * A = [inp | h] tile in LDS, row stride 2H+4 floats: one ds_read_b128 per lane feeds four MFMA k-steps
//     (k = 8kb + 4*(lane>>5) + j), conflict-free (16-lane phase groups hit 16 distinct 4-bank slots);
//   * B = weights streamed from L2, pre-packed as Wp[k/8][col][(k>>2)&1][k&3] (ic3_policy_pack): one coalesced
//     16-byte load per lane per gate per 8 k, two register buffers refilled a full 32-MFMA block ahead;
//   * LDS budget 64 x (2H+4) floats = 66.5 KB: the encoder output is staged through the h half, parked in the
//     accumulators of the C product (which it initialises), and the communication tile takes the inp half.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "env_device.hpp"
#include "ic3_common.hpp"

namespace ic3 {

typedef float ps_f32x4 __attribute__((ext_vector_type(4)));
typedef float ps_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32, optionally with the accumulator pinned to AGPRs (IC3_PS_AGPR=1): hipcc picks the all-VGPR
// form when the registers fit, which streams ~6 % slower in isolation (144 vs 153 TFLOP/s, tools/exp/ws_probe.hip).
// The asm is opaque to the hazard recogniser: whoever reads the accumulator afterwards calls mfma_settle() first.
// -DIC3_PS_TRACE: wave 0 of every workgroup stamps s_memrealtime (100 MHz) at the phase boundaries into a device buffer;
// the 40th ic3_policy_step call of the process dumps it to $IC3_PS_TRACE_OUT (tools/build_variant.sh trace -DIC3_PS_TRACE)
#ifdef IC3_PS_TRACE
#define IC3_TR(k)                                                                                       \
    do {                                                                                                \
        if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 20 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define IC3_TR(k) do { } while (0)
#endif
#ifndef IC3_PS_ENC_UNROLL
#define IC3_PS_ENC_UNROLL 2   // rows of the sparse encoder gather in flight per thread
#endif
#ifndef IC3_PS_AGPR
#define IC3_PS_AGPR 0   // measured: the 128/128 VGPR/AGPR split spills in the phases around the loops; net slower (0.57 vs 0.52 ms)
#endif
__device__ __forceinline__ void mfma_acc(ps_f32x16& acc, float x, float y)
{
#if IC3_PS_AGPR
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ void mfma_settle()
{
#if IC3_PS_AGPR
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}

struct StepArgs {
    // policy (ic3_policy)
    const ps_f32x4* Wt;         // encoder.weight^T [obs_dim][H/4]
    const ps_f32x4* enc_bias;   // encoder.bias + C.bias [H/4]
    const ps_f32x4* loc_table;  // ic3_env_encode_table or null
    const float* enc_in;        // KIND 0 (no env): encoder(x) + C.bias computed by the caller, [R][H]
    const ps_f32x4* c_wp;       // packed C.weight
    const ps_f32x4* l_wp;       // packed [W_ih | W_hh]
    const float* l_bias;        // b_ih + b_hh [4H]
    const float* head_w;        // [OT][H]  heads then value head
    const float* head_b;        // [OT]
    int OT, nheads, a0, a1, a2, a3;
    int mode_avg, comm_zero;
    int zmode;                  // experiment bits (+4 / +8 / +16, see the kernel)
    unsigned long long* trace;  // IC3_PS_TRACE builds: [tiles][20] phase time stamps
    int zb, zl, zc;               // zero-store pacing: per burst in front of the gate loop; inside it one nibble per k sub-step
    int skew;                   // IC3_PS_SKEW: workgroups 256..511 start this many s_sleep(127) late (phase offset
                                // between the two co-resident workgroups of a CU; speed only)
    int dbg;                    // timing ablations (IC3_PS_DEBUG bit mask; results are wrong when set): 1 gate MFMA loop,
                                // 2 C product, 4 encoder gather, 8 heads / draws / env step, 16 epilogue HBM traffic, 32 obs patch pass
    // recurrent state, masks, outputs
    float* h;                   // [R][H] in place
    float* c;                   // [R][H] in place
    const int32_t* alive_in;    // [R] or null (t = 0: everyone alive, quirk Q21)
    const int32_t* comm_in;     // [R] or null (gate sampled at t-1, quirk Q22)
    float* out;                 // [R][OT] log-probs | value
    int32_t* action;            // [nheads][R]
    float* obs;                 // [E][N][obs_dim] or null: next_state rows, stored from inside this kernel
    int obs_dim;                // floats per observation row
    int ntiles;                 // workgroups = tiles: n_full tiles of EPT envs, then half tiles of EPTh envs
    int n_full, EPTh;
    // env
    int E, N, EPT, G;
    int auto_reset;             // env handle in auto-reset mode: an env with t == 0 starts an episode (h = c = 0, gate 0)
    int tile_words;             // int32 words of one env-descriptor block in LDS
    uint32_t seed, gid0;
    const int32_t* episode;
    const int32_t* tstep;
    StepOut so;
    PPState pp;
    TJState tj;
};

template <int H, int KIND>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 2 : 1) void policy_step_kernel(const StepArgs a)
{
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, BM = 64, NT = 2 * H, NW = H / 32, H4 = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                            // [BM][LDA]: cols [0,H) inp / comm, [H,2H) h / enc / h'
    ps_f32x4* As4 = reinterpret_cast<ps_f32x4*>(smem);
    float* sm = As + BM * LDA;                                   // [BM] m_j = alive_j * comm_action_j
    float* sscale = sm + BM;                                     // [BM] per-env 1/(n_alive-1)
    int32_t* sact = reinterpret_cast<int32_t*>(sscale + BM);     // [BM] env action (head 0) of every row
    uint32_t* rmask = reinterpret_cast<uint32_t*>(sact + BM);    // [BM] window cells of every row that carry a count
    uint32_t* sfm = rmask + BM;                                  // [2] (+2 pad) bit r: row r starts an episode (auto-reset)
    int32_t* tile = reinterpret_cast<int32_t*>(sfm + 4);         // env descriptors of the tile's envs

    if (a.skew > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(127);
    const int tid = threadIdx.x;
    IC3_TR(0);
#ifdef IC3_PS_TRACE
    if (a.trace && tid == 0) {
        a.trace[(size_t)blockIdx.x * 20 + 18] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        a.trace[(size_t)blockIdx.x * 20 + 19] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
    const int N = a.N;
    const int WW = (KIND == 0) ? 0 : (KIND == IC3_ENV_PP) ? (2 * a.pp.v + 1) * (2 * a.pp.v + 1) : (2 * a.tj.v + 1) * (2 * a.tj.v + 1);
    const int total = a.pp.Np + a.pp.nprey;
    const int nsegE = N * WW;
    const int tjw = tj_tile_words(N, WW);
    const float inv_WW = 1.0f / (float)max(WW, 1);
    const float invN = 1.0f / (float)N, inv_nsegE = 1.0f / (float)max(nsegE, 1);   // div_small(): no integer divisions
    // env descriptors of `ne` envs starting at env `eb`, into the LDS block `tl` (two phases around a barrier):
    //   PP: sr[EPT*total] | sc[EPT*total] | tab[EPT*N*WW] (int2);  TJ: EPT x TJTile
    auto desc_positions = [&](int32_t* tl, int eb, int ne) {
        if constexpr (KIND == IC3_ENV_PP) {
            int32_t* psr = tl;
            int32_t* psc = tl + a.EPT * total;
            for (int i = tid; i < ne * total; i += NT) {
                psr[i] = a.pp.loc_r[(size_t)eb * total + i];
                psc[i] = a.pp.loc_c[(size_t)eb * total + i];
            }
        } else if constexpr (KIND == IC3_ENV_TJ) {
            for (int i = tid; i < ne * N; i += NT) {
                const int el = div_small(i, invN);
                tj_tile_load_car(tj_tile_at(tl + el * tjw, N), a.tj, eb + el, i - el * N);
            }
        }
    };
    auto desc_tab = [&](int32_t* tl, int ne) {
        if constexpr (KIND != 0) {
            int2* pt = reinterpret_cast<int2*>(tl + ((2 * a.EPT * total + 3) & ~3));
            for (int s = tid; s < ne * nsegE; s += NT) {
                const int el = div_small(s, inv_nsegE), q = s - el * nsegE;
                int2 d;
                if constexpr (KIND == IC3_ENV_PP) {
                    d = pp_tab_entry(tl + el * total, tl + a.EPT * total + el * total, q, a.pp.Np, total, a.pp.dim, a.pp.v);
                    pt[s] = d;
                } else {
                    const TJTile t = tj_tile_at(tl + el * tjw, N);
                    d = tj_tab_entry(t, a.tj, q);
                    t.tab[q] = d;
                }
                if (d.y != 0 && WW <= 32) {          // rows of the encoder only visit the cells flagged here
                    const int ag = div_small(q, inv_WW);
                    atomicOr(&rmask[el * N + ag], 1u << (q - ag * WW));
                }
            }
        }
    };
    // one workgroup per tile: the hardware dispatcher balances the tiles over the CUs (a resident set of workgroups
    // walking a strided tile list was measured slower: 350 vs 327 us, and needed tricks against hoisted loads)
    const int tile_id = blockIdx.x;
    constexpr int tz = 0;
    const int lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int col = 32 * w + li;
    // full tiles first; the envs left over behind the last round that gives every CU the same number of them go out as
    // HALF tiles (<= 32 rows: one 32-row MFMA tile, half the matrix work) — see plan_tiles()
    const bool half = tile_id >= a.n_full;
    const int e0 = half ? a.n_full * a.EPT + (tile_id - a.n_full) * a.EPTh : tile_id * a.EPT;
    const int nenv = min(half ? a.EPTh : a.EPT, a.E - e0);
    const int rows = nenv * N;                                   // valid rows of this tile (<= 64; <= 32 in a half tile)
    const bool two = rows > 32;                                  // second 32-row MFMA tile in use (workgroup-uniform)
    const size_t r0 = (size_t)e0 * N;

    // ---- dense observation of the state this step acts on (the `state` the reference hands to policy_net,
    // trainer.py:49), written by the launch that consumes it.  A wave that streams fp32 MFMAs leaves no issue slots to
    // any other wave of its SIMD (measured: tools/exp/ws_probe.hip), so the store stream can only share time with the
    // matrix work from INSIDE the same instruction stream — and there every instruction counts.  The rows are ~98 %
    // zeros: the tile's contiguous slice of the obs tensor is ZERO-FILLED by stores sprinkled between the MFMAs of the
    // gate loop and the transcendentals of the LSTM epilogue (scalar bookkeeping only), and the few non-zero entries
    // (<= 3 per window cell) are patched in at the very end, after every wave has seen its zero stores complete
    // (s_waitcnt + barrier).
    const bool obs_here = (KIND != 0) && a.obs != nullptr;
    const long long ob0 = (long long)e0 * N * a.obs_dim;         // first float of the tile's rows
    const int oL = rows * a.obs_dim;                             // floats of the tile
    const int ohead = (int)((4 - (ob0 & 3)) & 3);
    const int onb = obs_here ? (oL - ohead) >> 2 : 0;            // float4s of the body
    ps_f32x4* const obody = reinterpret_cast<ps_f32x4*>(a.obs + ob0 + ohead);
    // The body is cut into 1 KiB-aligned chunks of 64 float4s (see pp_obs_kernel); chunk c holds body indices
    // [64c - mis, 64c - mis + 64).  The (at most two) ragged chunks at the ends go out here with lane predicates; the
    // full ones are dealt round-robin to the waves and issued by zero_store() with wave-uniform control only: a scalar
    // count, a scalar base address (SGPR pair, bumped by scalar adds), one constant lane offset and a zero vector held
    // in registers — no vector ALU work, no exec masking, nothing for the matrix pipe to wait for.
    // Cache policy: non-temporal.  1.2 GB of zeros per launch flow through the 4 MB L2s next to the 0.6 MB of weights
    // every tile streams from there: with plain stores (a -DIC3_PS_PLAIN_STORES build) the kernel takes 0.50 ms
    // instead of 0.38.
    // (zmode +16: no L2 warm-up of c; +32: rest of the zero fill right behind the loop — experiments)
    const int mis = (int)(((ob0 + ohead) >> 2) & 63);
    const int c_lo = mis ? 1 : 0, c_hi = (mis + onb) >> 6;       // full chunks: [c_lo, c_hi)
    const int ws = __builtin_amdgcn_readfirstlane(tid >> 6);
    int zleft = 0;
    uint32_t zb_lo = 0, zb_hi = 0;
    if (obs_here) {
        zleft = __builtin_amdgcn_readfirstlane(max(0, (c_hi - c_lo - ws + NW - 1) / NW));
        const uint64_t zb = (uint64_t)obody + (uint64_t)((long long)(64 * (c_lo + ws) - mis) * 16);
        zb_lo = __builtin_amdgcn_readfirstlane((uint32_t)zb);
        zb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(zb >> 32));
    }
    const uint32_t zoff = (uint32_t)lane * 16u;
    ps_f32x4 zv = { 0.f, 0.f, 0.f, 0.f };
    asm volatile("" : "+v"(zv));                                 // keep it in registers (no re-materialisation per store)
    auto zero_store = [&]() {
        if (zleft > 0) {
            const uint64_t zb = ((uint64_t)zb_hi << 32) | zb_lo;
#ifdef IC3_PS_PLAIN_STORES
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(zoff), "v"(zv), "s"(zb) : "memory");
#else
            asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(zoff), "v"(zv), "s"(zb) : "memory");
#endif
            const uint64_t nb = zb + (uint64_t)NW * 1024u;
            zb_lo = (uint32_t)nb;
            zb_hi = (uint32_t)(nb >> 32);
            --zleft;
        }
    };
    auto zero_burst = [&](int n) {
#pragma unroll 1
        for (int i = 0; i < n; ++i) zero_store();
    };
    if (obs_here) {
        // ragged chunks: chunk 0 when the body starts inside it, chunk c_hi when the body ends inside it
        const int q0 = lane - mis, q1 = 64 * c_hi - mis + lane;
        if (ws == 0 && mis && q0 >= 0 && q0 < onb) obody[q0] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
        if (ws == 1 % NW && ((mis + onb) & 63) && (c_hi > 0 || !mis) && q1 >= 0 && q1 < onb)
            obody[q1] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
    }
    if (obs_here) {
        const int otail = (oL - ohead) & 3;
        if (tid < ohead) a.obs[ob0 + tid] = 0.f;
        if (tid < otail) a.obs[ob0 + ohead + 4 * (long long)onb + tid] = 0.f;
    }

    // ---- S0: masks, per-env scale (comm.py:102-107,194-196; quirks Q21/Q23), entity positions --------------------
    // auto-reset: an env whose t == 0 is at the start of an episode — no alive mask yet (everyone counts as alive,
    // quirk Q21), gate 0 (no communication on the first step, quirk Q22), zero LSTM state (trainer.py:38-51)
    const bool autor = (KIND != 0) && a.auto_reset;              // workgroup-uniform
    auto fresh_row = [&](int row) {                              // only called when autor
        const int el = (int)(((float)row + 0.5f) * (1.0f / (float)N));   // row / N, exact for row < 64
        return a.tstep[e0 + el] == 0;
    };
    for (int r = tid; r < BM; r += NT) {                        // (NT >= 128: this is exactly wave 0, all lanes)
        float m = 0.f;
        const bool fr = autor && r < rows && fresh_row(r);
        if (r < rows && !fr)
            m = (float)((a.alive_in ? a.alive_in[r0 + r] : 1) * (a.comm_in ? a.comm_in[r0 + r] : 1));
        sm[r] = m;
        if constexpr (KIND != 0) rmask[r] = (WW <= 32) ? 0u : ~0u;   // filled next to the window descriptors (S1)
        if (autor) {                                              // one global read per row, here; later phases test a bit
            const unsigned long long fb = __ballot(fr);
            if (lane == 0) {
                sfm[0] = (uint32_t)fb;
                sfm[1] = (uint32_t)(fb >> 32);
            }
        }
    }
    for (int el = tid; el < nenv; el += NT) {
        int n_alive = 0;
        const bool fr = autor && a.tstep[e0 + el] == 0;
        for (int j = 0; j < N; ++j) n_alive += (a.alive_in && !fr) ? a.alive_in[r0 + (size_t)el * N + j] : 1;
        sscale[el] = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
    }
    int32_t* sr = tile;
    int32_t* sc = tile + a.EPT * total;
    int2* ptab = reinterpret_cast<int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
    desc_positions(tile, e0, nenv);
    // h rows of the tile: requested now (HBM latency runs under S1/S2), parked in registers until the encoder output
    // has left the h half of the LDS tile
    ps_f32x4 hv[8];
    {
        // rows are contiguous: float4 number idx of the tile sits at byte 16 * idx; rows >= `rows` read as zeros
        // (descriptor range check), the constant part of the offset rides on the scalar operand
        const __amdgpu_buffer_rsrc_t rhh = __builtin_amdgcn_make_buffer_rsrc(
            static_cast<void*>(a.h + r0 * H), 0, (uint32_t)rows * H * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            hv[i] = __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rhh, tid * 16, i * NT * 16, 0));
        }
    }
    // (a few zero stores per burst in front of the gate loop, see the pacing notes in ic3_policy_step)
    zero_burst(a.zb);
    __syncthreads();
    IC3_TR(1);
    unsigned long long fmask = 0;                                // rows that start an episode: zero h / c, no masks
    if (autor)
        fmask = (unsigned long long)__builtin_amdgcn_readfirstlane(sfm[0]) |
                ((unsigned long long)__builtin_amdgcn_readfirstlane(sfm[1]) << 32);

    // ---- S1: window descriptors ------------------------------------------------------------------------------------
    if constexpr (KIND != 0) {
        desc_tab(tile, nenv);
        __syncthreads();
        IC3_TR(2);
    }
    zero_burst(a.zb);
    // encoder weight rows / pre-summed location rows behind buffer descriptors (32-bit gather offsets)
    const BufRows encW = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.Wt), 0,
                                                             (uint32_t)((size_t)a.obs_dim * H * sizeof(float)), 0x00020000),
                           a.Wt != nullptr };
    const BufRows encL = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.loc_table), 0, 0x7fffffffu, 0x00020000),
                           a.loc_table != nullptr };
    // ---- S2: encoder(obs) + C.bias as a sparse gather (comm.py:51,119; pp/tj_encode_kernel) -> h half of the tile ----
#pragma unroll IC3_PS_ENC_UNROLL
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / H4, c4 = idx - row * H4;
        ps_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
        if (row < rows && !(a.dbg & 4)) {
            const int el = div_small(row, invN), aa = row - el * N;
            if constexpr (KIND == 0) {
                v = *reinterpret_cast<const ps_f32x4*>(a.enc_in + (r0 + row) * H + 4 * c4);
            } else if constexpr (KIND == IC3_ENV_PP) {
                v = pp_encode_row_t(sr + el * total, sc + el * total, ptab + el * nsegE, aa, c4, H4, WW,
                                    a.pp.dim * a.pp.dim + 4, a.pp.dim, encW, a.enc_bias + tz, encL, rmask[row]);
            } else {
                v = tj_encode_row_t(tj_tile_at(tile + el * tjw, N), a.tj, aa, c4, H4, encW, a.enc_bias + tz, encL,
                                    rmask[row]);
            }
        }
        As4[row * LDA4 + H4 + c4] = v;
    }
    __syncthreads();
    IC3_TR(3);

    // ---- S3: the encoder output moves into the accumulators of the C product (MFMA C/D layout:
    //      col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----------------------------------------------------
    ps_f32x16 accC[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt == 1 && !two) break;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            accC[rt][reg] = As[lr * LDA + H + col];
        }
    }
    __syncthreads();
    IC3_TR(4);

    zero_burst(a.zb);
    // ---- S4: h -> h half ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / H4, c4 = idx - row * H4;
        As4[row * LDA4 + H4 + c4] = (autor && ((fmask >> row) & 1)) ? ps_f32x4{ 0.f, 0.f, 0.f, 0.f } : hv[i];
    }
    __syncthreads();
    IC3_TR(5);

    zero_burst(a.zb);
    if (!a.comm_zero) {   // comm_mask_zero (comm.py:40-41): C sees zeros, inp = enc + C.bias
        // ---- S5: comm_j = m_j (S_e - m_j h_j) scale_e (closed form of comm.py:181-205) -> inp half --------------------
        {
            const int c4 = tid % H4;
            for (int el = tid / H4; el < nenv; el += NT / H4) {
                const ps_f32x4* hp = As4 + (el * N) * LDA4 + H4 + c4;
                const float scl = sscale[el];
                ps_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
                for (int i = 0; i < N; ++i) S += sm[el * N + i] * hp[i * LDA4];
                for (int j = 0; j < N; ++j) {
                    const float m = sm[el * N + j];
                    As4[(el * N + j) * LDA4 + c4] = m * (S - m * hp[j * LDA4]) * scl;
                }
            }
            for (int idx = rows * H4 + tid; idx < BM * H4; idx += NT) {
                const int row = idx / H4, c4p = idx - row * H4;
                As4[row * LDA4 + c4p] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
        }
        // B fragments of C: lane (li, lh) of wave w reads Wp[kb][32w + li][lh] -> k = 8kb + 4lh + j, j = 0..3
        constexpr int KBC = H / 8, CH = (KBC < 8) ? KBC : 8, NCH = KBC / CH;
        // weights through a buffer descriptor: lane offset in one VGPR, the k / gate part of the address on the scalar ALU
        const __amdgpu_buffer_rsrc_t rcw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<ps_f32x4*>(a.c_wp), 0, (uint32_t)((size_t)H * H * sizeof(float)), 0x00020000);
        const int wlane = (col * 2 + lh) * 16;
        auto cwp = [&](int k) {
            return __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rcw, wlane, k * (H * 2 * 16), 0));
        };
        ps_f32x4 cb[2][CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) cb[0][k] = cwp(k);
        __syncthreads();
        IC3_TR(6);
        // ---- S6: accC (= enc) += comm . C.weight^T ---------------------------------------------------------------------
        auto cprod = [&](auto two_c) {
            constexpr bool TWO = decltype(two_c)::value;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) cb[(ch + 1) & 1][k] = cwp((ch + 1) * CH + k);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int kb = ch * CH + k;
                    const ps_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
                    ps_f32x4 a1;
                    if constexpr (TWO) a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mfma_acc(accC[0], a0[j], cb[ch & 1][k][j]);
                        if constexpr (TWO) mfma_acc(accC[1], a1[j], cb[ch & 1][k][j]);
                    }
                    if (obs_here && kb < a.zc) zero_store();
                }
            }
        };
        if (!(a.dbg & 2)) {
            if (two) cprod(std::true_type{});
            else cprod(std::false_type{});
        }
        mfma_settle();
        __syncthreads();   // every wave has read the comm tile
        IC3_TR(7);
    }

    // gate weights: the first two 8-k blocks are requested before inp is written back
    constexpr int KB = K / 8;
    constexpr size_t KB_STRIDE = (size_t)4 * H * 2;   // float4s per kb
    const __amdgpu_buffer_rsrc_t rgw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ps_f32x4*>(a.l_wp), 0, (uint32_t)((size_t)K * 4 * H * sizeof(float)), 0x00020000);
    const int glane = (col * 2 + lh) * 16;
    auto wp = [&](int kb, int g) {   // float4 of gate g, k block kb (KB_STRIDE float4s per block, 2 H per gate)
        return __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rgw, glane, kb * (int)(KB_STRIDE * 16) + g * (H * 2 * 16), 0));
    };
    // (same issue order as inside the loop — all of b0, then all of b1 — so that the s_waitcnt vmcnt(n) the compiler
    // places in front of each MFMA group count exactly the loads that group needs on both paths into the loop)
    ps_f32x4 b0[4], b1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b0[g] = wp(0, g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) b1[g] = wp(1, g);
    __builtin_amdgcn_sched_barrier(0);
    zero_burst(a.zb);
    // ---- S7: inp = enc + C.bias + C(comm) -> inp half ------------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt == 1 && !two) break;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            As[lr * LDA + col] = accC[rt][reg];
        }
    }
    __syncthreads();
    IC3_TR(8);

    // ---- S8: gates = [inp | h] . [W_ih | W_hh]^T (comm.py:215, torch.nn.LSTMCell) --------------------------------------
    ps_f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][g][i] = 0.0f;
    auto block = [&](auto two_c, const ps_f32x4 (&bq)[4], int kb) {
        constexpr bool TWO = decltype(two_c)::value;
        const ps_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
        ps_f32x4 a1;
        if constexpr (TWO) a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nz = (a.zl >> (4 * j)) & 15;               // wave-uniform, loop-invariant
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mfma_acc(acc[0][g], a0[j], bq[g][j]);
                if constexpr (TWO) mfma_acc(acc[1][g], a1[j], bq[g][j]);
                // the scalar bookkeeping of a store slot fits into the 64-cycle shadow of one MFMA: one slot after
                // the 4th and one after the 8th of a k sub-step rather than both at its end
                if (g == 1 && nz > 1) zero_store();
                if (g == 3 && nz > 0) zero_store();
            }
#pragma unroll 1
            for (int i = 2; i < nz; ++i) zero_store();
        }
    };
    float sink = 0.0f;   // destination of the L2 warm-up load of c (see below)
    auto gate_loop = [&](auto two_c) {
        static_assert(KB % 2 == 0 && KB >= 4, "K/8 must be even");
        // sched_barrier(0) pins the phase order (the machine scheduler otherwise sinks the refill loads to just before
        // their first use, which exposes the full L2 latency every block).
        const int kb_end = (a.dbg & 1) ? 0 : KB - 2;
    #pragma unroll 1
        for (int kb = 0; kb < kb_end; kb += 2) {
            block(two_c, b0, kb);
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int g = 0; g < 4; ++g) b0[g] = wp(kb + 2, g);
            __builtin_amdgcn_sched_barrier(0);
            block(two_c, b1, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int g = 0; g < 4; ++g) b1[g] = wp(kb + 3, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the old cell state: one touch per 128 B line of this wave's (64 rows x 32 columns) before the last two blocks
        // (64 MFMAs) brings it from HBM into the L2 under them; the epilogue's loads then hit there.  (Holding the values
        // themselves over the two blocks costs 32 registers the loop does not have.)
        // The load's destination register stays reserved until the epilogue has waited for it (the compiler does not
        // know that an asm load completes later).
        const bool warm_c = !(a.zmode & 16) && lane < rows && !(a.dbg & 16);
        if (warm_c) {
            const float* cp = a.c + (r0 + lane) * H + 32 * w;
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(cp) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.dbg & 1)) {
            block(two_c, b0, KB - 2);
            __builtin_amdgcn_sched_barrier(0);
            block(two_c, b1, KB - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (two) gate_loop(std::true_type{});
    else gate_loop(std::false_type{});
    IC3_TR(9);
    mfma_settle();
    IC3_TR(10);
    if (a.zmode & 32)
        while (zleft > 0) zero_store();   // (experiment: the rest right behind the loop instead of inside the epilogue)
    // ---- S9: LSTM cell epilogue (gate order i,f,g,o); c', h' to HBM, h' also into the h half for the heads ------------
    {
        const float* lb = a.l_bias + tz;
        const float bi = lb[col], bf = lb[H + col], bg = lb[2 * H + col], bo = lb[3 * H + col];
        // c / h rows of the tile through buffer descriptors: one 32-bit lane offset + a constant per element instead of
        // a 64-bit address pair each, and the hardware range check (num_records = the tile's valid rows) stands in
        // for the `row < rows` predicates — an out-of-range load returns 0, an out-of-range store is dropped.
        const uint32_t nrec = (a.dbg & 16) ? 0u : (uint32_t)rows * H * 4u;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.c + r0 * H), 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.h + r0 * H), 0, nrec, 0x00020000);
        const int voff = (4 * lh * H + col) * 4;
        float cold[2][16];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt == 1 && !two) break;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                cold[rt][reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, voff + lc * H * 4, 0, 0));
                if (autor && ((fmask >> (lc + 4 * lh)) & 1)) cold[rt][reg] = 0.0f;
            }
        }
        __syncthreads();   // every wave is done with the A tile
        IC3_TR(11);
        for (int i = tid; i < a.OT * H4; i += NT) {   // head / value weights -> rows [0, OT) of the inp half
            const int o = i / H4, c4 = i - o * H4;
            As4[o * LDA4 + c4] = reinterpret_cast<const ps_f32x4*>(a.head_w)[i + tz];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt == 1 && !two) break;              // half tile: rows 32..63 are padding (their h' is never read)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                const float gi = acc[rt][0][reg] + bi, gf = acc[rt][1][reg] + bf;
                const float gg = acc[rt][2][reg] + bg, go = acc[rt][3][reg] + bo;
                const float c1 = fast_sigmoid(gf) * cold[rt][reg] + fast_sigmoid(gi) * fast_tanh(gg);
                const float h1 = fast_sigmoid(go) * fast_tanh(c1);
                zero_store();                      // what the gate loop left of the zero fill goes out between the
                zero_store();                      // transcendental work of the cell (2 x 32 slots, then the rest)
                const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c1), rc, voff + lc * H * 4, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, h1), rh, voff + lc * H * 4, 0, 0);
                As[lr * LDA + H + col] = h1;
            }
        }
        while (zleft > 0) zero_store();            // obs-dominated shapes
        // the warm-up load's destination stayed reserved up to here: memory operations complete in order, so it landed
        // before the first cold[] value (requested after it) was consumed
        asm volatile("" : : "v"(sink));
    }
    __syncthreads();
    IC3_TR(12);
    if (a.dbg & 8) return;

    // ---- S10: heads + value head (comm.py:228,239) as a 64 x 16 x H product on v_mfma_f32_16x16x4_f32: row tile of 16
    //      rows per wave, the OT <= 16 output columns are the weight rows [0, 16) of the inp half (rows >= OT hold
    //      stale finite data and only feed output columns nobody reads).  Operand layout of the instruction: A[i][k] in
    //      lane 16k + i, B[k][j] in lane 16k + j, D[4(l/16) + v][l % 16] in element v of lane l; one ds_read_b128 per
    //      operand feeds four k-steps (k = 16 sg + 4 (l/16) + j — any k order is valid as long as A and B agree).
    // logits of row r -> rows [16, ..) of the inp half: z(r, o) = As[(16 + r / PER) * LDA + (r % PER) * 16 + o]
    constexpr int PER = H / 16;
    {
        const int l16 = lane & 15, kq = lane >> 4;
        const float hb = l16 < a.OT ? a.head_b[l16 + tz] : 0.0f;
        for (int rtile = w; rtile < BM / 16; rtile += NW) {
            if (16 * rtile >= rows) break;
            ps_f32x4 z = { 0.f, 0.f, 0.f, 0.f };
            const ps_f32x4* xa = As4 + (16 * rtile + l16) * LDA4 + H4 + kq;
            const ps_f32x4* wb = As4 + l16 * LDA4 + kq;
#pragma unroll 4
            for (int sg = 0; sg < H / 16; ++sg) {
                const ps_f32x4 x4 = xa[4 * sg], w4 = wb[4 * sg];
#pragma unroll
                for (int j = 0; j < 4; ++j) z = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[j], w4[j], z, 0, 0, 0);
            }
            if (l16 < a.OT) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * rtile + 4 * kq + v;
                    As[(16 + r / PER) * LDA + (r % PER) * 16 + l16] = z[v] + hb;
                }
            }
        }
    }
    __syncthreads();
    IC3_TR(13);

    // ---- S11: log_softmax per head + the action draws (action_utils.py:32-36; same arithmetic and Philox counters as
    //      lstm_cell_heads_kernel / sample_actions_env_kernel), one task per (row, head) + one per row for the value ----
    {
        const int sizes[4] = { a.a0, a.a1, a.a2, a.a3 };
        const int R = a.E * N;
        const float inv_nh1 = 1.0f / (float)(a.nheads + 1);
        for (int task = tid; task < rows * (a.nheads + 1); task += NT) {
            const int tr = div_small(task, inv_nh1), hd = task - tr * (a.nheads + 1);
            const size_t grow = r0 + tr;
            const float* z = As + (16 + tr / PER) * LDA + (tr % PER) * 16;
            float* orow = a.out + grow * a.OT;
            int off = 0;
            for (int i = 0; i < hd && i < a.nheads; ++i) off += sizes[i];
            if (hd == a.nheads) {                   // value head (last column)
                orow[off] = z[off];
                continue;
            }
            const int A = sizes[hd];
            float mx = -INFINITY;
            for (int o = 0; o < A; ++o) mx = fmaxf(mx, z[off + o]);
            float sum = 0.0f;                       // hardware exp2 / log2 (~1 ulp): |error| of a log-prob ~1e-7, bar 1e-5
            for (int o = 0; o < A; ++o) sum += __builtin_amdgcn_exp2f(1.4426950408889634f * (z[off + o] - mx));
            const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(sum);
            for (int o = 0; o < A; ++o) orow[off + o] = z[off + o] - lse;
            if (KIND == 0) continue;                // forward only: the caller draws (ic3_sample_actions)
            const int el = div_small(tr, invN), n = tr - el * N;
            const int e = e0 + el;
            const uint32_t x = philox_x24(a.seed, a.gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)a.episode[e],
                                          (uint32_t)a.tstep[e], (uint32_t)(hd * N + n));
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
            a.action[(size_t)hd * R + grow] = act;
            if (hd == 0) sact[tr] = act;
        }
    }
    __syncthreads();
    IC3_TR(14);

    // ---- S12: env.step for the tile's envs with the env-action head (env_wrappers.py:76-77) ----------------------------
    if constexpr (KIND != 0) {
        const int lgG = __builtin_ctz(a.G);
        for (int base = 0; base < a.EPT * a.G; base += NT) {
            const int lt = base + tid;
            const int el = lt >> lgG, n = lt - (el << lgG);         // G is a power of two
            const int e = el < nenv ? e0 + el : a.E;
            if constexpr (KIND == IC3_ENV_PP) {
                pp_step_lanes(a.pp, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
            } else {
                tj_step_lanes(a.tj, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
            }
        }
    }
    IC3_TR(15);
    if (obs_here && !(a.dbg & 32)) {
        // every zero store of this workgroup has completed (own stores: vmcnt(0); the others': barrier) before the
        // first non-zero entry goes out to the same lines
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        IC3_TR(16);
        float* orow0 = a.obs + ob0;
        if constexpr (KIND == IC3_ENV_PP) {
            const int vocab = a.pp.dim * a.pp.dim + 4;
            for (int sg = tid; sg < nenv * nsegE; sg += NT) {   // descriptors of the INPUT state (S1)
                const int2 d = ptab[sg];
                float* cell = orow0 + (size_t)sg * vocab;
                const float npred = (float)(d.y & 0xffff), nprey = (float)(d.y >> 16);
                // channels: d.x one-hot (grid id or OUTSIDE), vocab-2 #prey, vocab-1 #predators (counts add, quirk Q3)
                cell[d.x] = 1.f + (d.x == vocab - 2 ? nprey : 0.f) + (d.x == vocab - 1 ? npred : 0.f);
                if (d.x != vocab - 2 && nprey != 0.f) cell[vocab - 2] = nprey;
                if (d.x != vocab - 1 && npred != 0.f) cell[vocab - 1] = npred;
            }
        } else if constexpr (KIND == IC3_ENV_TJ) {
            const int obs_dim = a.obs_dim;
            for (int sg = tid; sg < nenv * (nsegE + N); sg += NT) {
                const int el = div_small(sg, 1.0f / (float)(nsegE + N)), q = sg - el * (nsegE + N);
                tj_obs_patch(tj_tile_at(tile + el * tjw, N), a.tj, orow0 + (size_t)el * N * obs_dim, obs_dim, WW, q);
            }
        }
    }
    IC3_TR(17);
}

// Wp[kb][col][hh][j] = W[col][8 kb + 4 hh + j], W = [Wa | Wb] (C x (Ka + Kb)) row-major halves
__global__ void policy_pack_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb, float* __restrict__ Wp,
                                   int C, int Ka, int Kb)
{
    const int Kt = Ka + Kb;
    const long long n = (long long)C * Kt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3), hh = (int)((i >> 2) & 1);
        const long long rest = i >> 3;
        const int colx = (int)(rest % C), kb = (int)(rest / C);
        const int k = 8 * kb + 4 * hh + j;
        Wp[i] = k < Ka ? Wa[(size_t)colx * Ka + k] : Wb[(size_t)colx * Kb + (k - Ka)];
    }
}

static int resident_workgroups(int H)
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 512;
        cus = prop.multiProcessorCount;
    }
    return cus * (H <= 128 ? 2 : 1);
}

// Tile plan (a.E, a.N, a.EPT set).  Two workgroups share a CU and all tiles cost the same, so a launch whose tile count
// is not a multiple of the slot count ends with a round in which some CUs still hold two tiles while others hold one
// or none (PP-hard: 1366 tiles = 2.67 rounds of 512 slots cost 3).  Plan B: as many FULL tiles (EPT envs, two 32-row
// MFMA tiles) as give every CU the same number, the rest as HALF tiles (EPTh = floor(32 / N) envs, one MFMA tile) that
// are dispatched last and land next to a CU's last full tile (or alone).  A half tile is not half the time — the phases
// around the MFMA loops and the weight stream stay — so plan B is chosen only when a cost model calibrated on PP-hard /
// TJ-hard / E = 384 says it ends earlier: pair of full tiles 1.0, full + half 0.91, pair of halves 0.70, lone full
// 0.6, lone half 0.47 (PP-hard 0.321 -> 0.311 ms, TJ-medium 0.297 -> 0.290; TJ-hard and PP-easy stay with plan A).
// IC3_PS_HALF=0 / 1 forces plan A / B.
static double tiles_cost(int k_full, int k_half)
{
    double c = (k_full / 2) * 1.0;
    if (k_full & 1) {
        if (k_half > 0) {
            c += 0.91;
            --k_half;
        } else {
            c += 0.6;
        }
    }
    return c + (k_half / 2) * 0.70 + (k_half & 1) * 0.47;
}

static int plan_tiles(StepArgs& a, int H)
{
    static const int force = getenv("IC3_PS_HALF") ? atoi(getenv("IC3_PS_HALF")) : -1;
    const int cus = resident_workgroups(H) / (H <= 128 ? 2 : 1);
    const int n_all = (a.E + a.EPT - 1) / a.EPT;
    a.EPTh = 32 / a.N;
    a.n_full = n_all;
    a.ntiles = n_all;
    if (a.EPTh < 1 || H > 128 || force == 0) return a.ntiles;
    const int n_full = (a.E / a.EPT) / cus * cus;               // every CU the same number of full tiles
    const int rem = a.E - n_full * a.EPT;
    const int n_half = (rem + a.EPTh - 1) / a.EPTh;
    const double cost_a = tiles_cost((n_all + cus - 1) / cus, 0);
    const double cost_b = tiles_cost(n_full / cus, (n_half + cus - 1) / cus);
    if (force == 1 || cost_b < cost_a - 1e-9) {
        a.n_full = n_full;
        a.ntiles = n_full + n_half;
    }
    return a.ntiles;
}

template <int H, int KIND>
static int launch_step(const StepArgs& a, int tiles, size_t lds, hipStream_t s, hipEvent_t ev0 = nullptr,
                       hipEvent_t ev1 = nullptr)
{
    static bool attr_set = false;
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && (!attr_set || lds > attr_lds)) {
        IC3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_step_kernel<H, KIND>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
        attr_lds = lds;
    }
    // IC3_PS_WGS=1: ask for more than half of the CU's LDS so that only ONE workgroup is resident per CU (experiments
    // with a concurrent obs-assembly launch on a second stream, which then finds free wave slots and registers)
    static const int one_wg = getenv("IC3_PS_WGS") ? atoi(getenv("IC3_PS_WGS")) == 1 : 0;
    if (one_wg && lds < 84 * 1024) {
        lds = 84 * 1024;
        if (!attr_set || lds > attr_lds) {
            IC3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_step_kernel<H, KIND>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
            attr_lds = lds;
        }
    }
    // one workgroup per tile, dispatched in tile order (full tiles first, see plan_tiles): the hardware dispatcher
    // balances them over the CUs (a fixed resident set walking a strided tile list was measured slower)
    const int grid = tiles;
    if (ev0 || ev1) {   // timed launch: the dispatch itself stamps the events (no separate record packets around it)
        hipExtLaunchKernelGGL((policy_step_kernel<H, KIND>), dim3(grid), dim3(2 * H), lds, s, ev0, ev1, 0, a);
    } else {
        hipLaunchKernelGGL((policy_step_kernel<H, KIND>), dim3(grid), dim3(2 * H), lds, s, a);
    }
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // namespace ic3

using namespace ic3;

extern "C" int ic3_policy_pack(const float* c_weight, const float* w_ih, const float* w_hh, float* c_wp, float* lstm_wp,
                               int H, ic3_stream stream)
{
    if (!c_weight || !w_ih || !w_hh || !c_wp || !lstm_wp || H <= 0 || (H & 31))
        return fail(-22, "ic3_policy_pack: H must be a positive multiple of 32");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(policy_pack_kernel, dim3(64), dim3(256), 0, s, c_weight, (const float*)nullptr, c_wp, H, H, 0);
    hipLaunchKernelGGL(policy_pack_kernel, dim3(256), dim3(256), 0, s, w_ih, w_hh, lstm_wp, 4 * H, H, H);
    IC3_HIP(hipGetLastError());
    return 0;
}

// LDS bytes of one workgroup (0 = unsupported shape); *tile_words_out = int32 words of one env-descriptor block
static int policy_step_lds(const ic3_env* env, int H, int with_obs, int* tile_words_out)
{
    if (!env) return 0;
    if (H != 64 && H != 128 && H != 256) return 0;
    const int N = env->dims.N;
    if (N < 1 || N > 64) return 0;
    const int EPT = 64 / N;
    const int WW = env->dims.window * env->dims.window;
    size_t tile_words;
    if (env->kind == IC3_ENV_PP) {
        const int total = env->pp.N + env->pp.nprey;
        tile_words = (size_t)((2 * EPT * total + 3) & ~3) + (size_t)2 * EPT * N * WW;
    } else {
        tile_words = (size_t)EPT * (((7 * N + 3) & ~3) + 2 * N * WW);
    }
    tile_words = (tile_words + 3) & ~(size_t)3;
    if (tile_words_out) *tile_words_out = (int)tile_words;
    (void)with_obs;
    const size_t lds = ((size_t)64 * (2 * H + 4) + 4 * 64 + 4 + tile_words) * sizeof(float);
    const size_t limit = (H <= 128) ? 80 * 1024 : 160 * 1024;   // two workgroups per CU up to H = 128
    return lds <= limit ? (int)lds : 0;
}

extern "C" int ic3_policy_step_supported(const ic3_env* env, int H) { return policy_step_lds(env, H, 0, nullptr); }

static int fill_policy(StepArgs& a, const ic3_policy* p, const char* who)
{
    if (!p->c_wp || !p->lstm_wp || !p->lstm_bias || !p->head_w || !p->head_b)
        return fail(-22, std::string(who) + ": incomplete ic3_policy");
    if (p->nheads < 1 || p->nheads > 4) return fail(-22, std::string(who) + ": 1..4 action heads");
    a.Wt = reinterpret_cast<const ps_f32x4*>(p->enc_wt);
    a.enc_bias = reinterpret_cast<const ps_f32x4*>(p->enc_bias);
    a.loc_table = reinterpret_cast<const ps_f32x4*>(p->loc_table);
    a.c_wp = reinterpret_cast<const ps_f32x4*>(p->c_wp);
    a.l_wp = reinterpret_cast<const ps_f32x4*>(p->lstm_wp);
    a.l_bias = p->lstm_bias;
    a.head_w = p->head_w;
    a.head_b = p->head_b;
    a.nheads = p->nheads;
    int sz[4] = { 0, 0, 0, 0 };
    a.OT = 1;
    for (int i = 0; i < p->nheads; ++i) {
        sz[i] = p->head_sizes[i];
        if (sz[i] < 1) return fail(-22, std::string(who) + ": empty action head");
        a.OT += sz[i];
    }
    if (a.OT > 16) return fail(-22, std::string(who) + ": more than 15 actions in total");
    a.a0 = sz[0];
    a.a1 = sz[1];
    a.a2 = sz[2];
    a.a3 = sz[3];
    a.mode_avg = p->mode_avg;
    a.comm_zero = p->comm_zero;
    static const int dbg = getenv("IC3_PS_DEBUG") ? atoi(getenv("IC3_PS_DEBUG")) : 0;
    a.dbg = dbg;
    static const int skew = getenv("IC3_PS_SKEW") ? atoi(getenv("IC3_PS_SKEW")) : 0;
    a.skew = skew;
    return 0;
}

extern "C" int ic3_policy_forward(const ic3_policy* p, const float* enc, int E, int N, float* h, float* c,
                                  const int32_t* alive_in, const int32_t* comm_in, float* out, ic3_stream stream)
{
    ic3::Range range_("ic3_policy_forward");
    if (!p || !enc || !h || !c || !out || E <= 0 || N <= 0) return fail(-22, "ic3_policy_forward: bad arguments");
    const int H = p->H;
    if ((H != 64 && H != 128 && H != 256) || N > 64)
        return fail(-38, "ic3_policy_forward: needs hid_size 64/128/256 and <= 64 agents per env");
    StepArgs a{};
    int rc = fill_policy(a, p, "ic3_policy_forward");
    if (rc) return rc;
    a.enc_in = enc;
    a.h = h;
    a.c = c;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.E = E;
    a.N = N;
    a.EPT = 64 / N;
    a.G = 1;
    const int tiles = plan_tiles(a, H);
    const size_t lds = ((size_t)64 * (2 * H + 4) + 4 * 64 + 4) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (H == 128) return launch_step<128, 0>(a, tiles, lds, s);
    if (H == 64) return launch_step<64, 0>(a, tiles, lds, s);
    return launch_step<256, 0>(a, tiles, lds, s);
}

extern "C" int ic3_policy_step(ic3_env* env, const ic3_policy* p, float* h, float* c, const int32_t* alive_in,
                               const int32_t* comm_in, float* out, int32_t* action, float* obs, float* reward,
                               int32_t* done, int32_t* alive, int32_t* is_completed, ic3_stream stream)
{
    ic3::Range range_("ic3_policy_step");
    if (!env || !p || !h || !c || !out || !action || !reward || !done)
        return fail(-22, "ic3_policy_step: null argument");
    if (env->resets == 0) return fail(-22, "ic3_policy_step: reset() has not been called");
    if (!p->enc_wt || !p->enc_bias) return fail(-22, "ic3_policy_step: incomplete ic3_policy (encoder)");
    const int H = p->H;
    // next_state rows are stored from inside the kernel when their descriptors fit in LDS next to the tile's own
    // (IC3_PS_OBS=0: always as a separate ic3_env_observe launch after the kernel)
    static const int obs_inside = getenv("IC3_PS_OBS") ? atoi(getenv("IC3_PS_OBS")) : 1;
    int tile_words = 0;
    int lds = (obs && obs_inside) ? policy_step_lds(env, H, 1, &tile_words) : 0;
    const bool fused_obs = lds != 0;
    if (!lds) lds = policy_step_lds(env, H, 0, &tile_words);
    if (!lds)
        return fail(-38, "ic3_policy_step: needs hid_size 64/128/256, <= 64 agents per env and an env tile that fits "
                         "in LDS (use ic3_env_encode + ic3_comm_masked_mean + GEMMs + ic3_lstm_cell_heads + ic3_env_step)");
    StepArgs a{};
    int frc = fill_policy(a, p, "ic3_policy_step");
    if (frc) return frc;
    a.h = h;
    a.c = c;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.action = action;
    a.E = env->dims.E;
    a.N = env->dims.N;
    a.EPT = 64 / a.N;
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
    const int tiles = plan_tiles(a, H);
    a.tile_words = tile_words;
    a.obs = fused_obs ? obs : nullptr;
    a.obs_dim = env->dims.obs_dim;
    a.auto_reset = env->auto_max_steps > 0;
    {   // pacing of the zero stores: per-thread count of a tile; KB*4 slots in the loop, then the cell epilogue
        static const int zmode_env = getenv("IC3_PS_ZMODE") ? atoi(getenv("IC3_PS_ZMODE")) : 0;
        a.zmode = zmode_env;
        static const int zb_env = getenv("IC3_PS_ZB") ? atoi(getenv("IC3_PS_ZB")) : -1;
        static const int zl_env = getenv("IC3_PS_ZL") ? (int)strtol(getenv("IC3_PS_ZL"), nullptr, 0) : -1;
        const long long per_thread = ((long long)a.EPT * a.N * a.obs_dim / 4 + 2 * H - 1) / (2 * H) + 1;
        const int slots = (2 * H / 8) * 4;                       // one per 8 MFMAs
        // zl: one nibble per k sub-step of a K block (8 MFMAs each) = zero stores issued after it.  None in front of
        // the loop: stores there delay the loads of the phases there (memory operations of a wave complete in
        // order), measured 0.439 -> 0.429 ms on PP-hard.
        static const int zc_env = getenv("IC3_PS_ZC") ? atoi(getenv("IC3_PS_ZC")) : -1;
        a.zc = zc_env >= 0 ? zc_env : 0;                         // inside the C product (one per 8 MFMAs): none, 0.4198 -> 0.4177 ms
        if (a.zc > H / 8) a.zc = H / 8;
        if (zl_env >= 0) {
            a.zl = zl_env;
        } else {
            // ~70 % of a tile's stores inside the loop (PP-hard: 5 of the 7.1 per K block, 0.397 -> 0.381 ms against
            // all of them), the rest inside the cell epilogue
            static const int zfrac = getenv("IC3_PS_ZFRAC") ? atoi(getenv("IC3_PS_ZFRAC")) : 70;
            long long want = ((per_thread - a.zc) * zfrac / 100 + slots / 8) / (slots / 4);   // per K block, rounded
            if (want > 60) want = 60;
            if (want < 0) want = 0;
            a.zl = 0;
            for (int j = 0; j < 4; ++j) a.zl |= (int)((want + 3 - j) / 4) << (4 * j);
        }
        // per burst between the five phases in front of the loop: 3 % of the tile's stores each (PP-hard: 7 of 228 per
        // thread; 0.327 -> 0.314 ms; more — or any, before the phases there lost their spills and scans — delays the
        // loads of those phases: memory operations of a wave complete in order)
        a.zb = zb_env >= 0 ? zb_env : (int)((per_thread * 3 + 50) / 100);
        if (a.zb > 48) a.zb = 48;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (obs && !fused_obs) {   // same contents, as a launch of its own in front (the step below changes the state)
        rc = ic3_env_observe(env, obs, stream);
        if (rc) return rc;
    }
#ifdef IC3_PS_TRACE
    static unsigned long long* trace_buf = nullptr;
    static int trace_call = 0, trace_tiles = 0;
    if (getenv("IC3_PS_TRACE_OUT")) {
        if (!trace_buf) {
            trace_tiles = tiles;
            IC3_HIP(hipMalloc(&trace_buf, (size_t)tiles * 20 * sizeof(unsigned long long)));
            IC3_HIP(hipMemset(trace_buf, 0, (size_t)tiles * 20 * sizeof(unsigned long long)));
        }
        if (tiles == trace_tiles) a.trace = trace_buf;
    }
#endif
    hipEvent_t ev0 = (hipEvent_t)env->ev_start, ev1 = (hipEvent_t)env->ev_stop;   // one-shot (ic3_env_set_step_events)
    env->ev_start = env->ev_stop = nullptr;
    if (H == 128)
        rc = pp ? launch_step<128, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<128, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
    else if (H == 64)
        rc = pp ? launch_step<64, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<64, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
    else
        rc = pp ? launch_step<256, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<256, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
#ifdef IC3_PS_TRACE
    if (a.trace && ++trace_call == 40) {
        IC3_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)trace_tiles * 20);
        IC3_HIP(hipMemcpy(h.data(), trace_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("IC3_PS_TRACE_OUT"), "w")) {
            for (int t = 0; t < trace_tiles; ++t) {
                fprintf(f, "%d", t);
                for (int k = 0; k < 20; ++k) fprintf(f, ",%llu", h[(size_t)t * 20 + k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
#endif
    return rc;
}
