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
//   * B = weights streamed from L2, pre-packed by ic3_policy_pack as Wq[k][c] = float4 over the four gates of hidden
//     column c: one coalesced 16-byte load per lane per k-step, a ring of four float4 refilled 24 MFMAs ahead
//     (the C product keeps Wp[k/8][col][(k>>2)&1][k&3]);
//   * LDS budget 64 x (2H+4) floats = 66.5 KB: the encoder output is staged through the h half, parked in the
//     accumulators of the C product (which it initialises), and the communication tile takes the inp half.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "env_device.hpp"
#include "ic3_common.hpp"
#include "ps_common.hpp"

namespace ic3 {

// Matrix instructions through the compiler's builtins only (an AGPR-pinned inline-asm form was measured in rounds 2-3: net
// slower here — the phases around the loops need more than 128 VGPRs — and opaque to the hazard recogniser).
//
// Wave priority: 3 in the phases in front of the gate loop (short dependent chains of loads, LDS exchanges and barriers
// whose every instruction is on the tile's critical path), 0 from the gate loop on — the co-resident workgroup's MFMA
// stream gives up an issue slot now and then, the phases stop queueing behind it (round 4, one box: TJ-hard +2.3 %,
// TJ-medium +1.7 %, PP-hard +0.6 %; also raising the epilogue or the phases behind it: no further gain; the reverse: -1.5 %).
// PRIO_AT(1) = front phases, (2) = cell epilogue, (4) = behind the epilogue.
#define IC3_PRIO_AT(bit) __builtin_amdgcn_s_setprio(((bit) & 1) ? 3 : 0)
// Cache policy of the split loop's weight fragment loads (default) and of the obs zero stores (nt: the rows are read by nobody
// on this chip before they fall out of L2; sc0 / sc1 variants and plain stores were measured in rounds 3-4: profiles/r03,
// profiles/r04/ab_runs.txt); float4 slots of the fp32 gate loop's B-operand ring (8 = two K blocks).
constexpr int PS_WLOAD_AUX = 0, PS_ZSTORE_AUX = 2, PS_RING = 8;

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
    // pacing of the obs zero fill (speed only — every store slot past the tile's slice is dropped by the hardware):
    int zs;                     // stores per K block of the gate loop (one of ZS_SET; a block = 32 MFMAs of a full tile)
    int zf, zepi, zh;           // stores in front of the comm phase; per element of the cell epilogue (0..2); in front of the heads
    int z0, z3, zc;             // stores behind the loads of S0; behind the request of the old cell state; K blocks of the C product with a slot
    int zrest;                  // stores per wave issued behind the cell epilogue (what the other slots left)
    // recurrent state, masks, outputs
    float* h;                   // [R][H] read ...
    float* c;                   // [R][H]
    float* h_out;               // ... and written here (= h, c unless ic3_env_set_hidden_out)
    float* c_out;
    float* gates_out;           // ic3_env_set_record_out: [R][4H] the activated gates i | f | g | o of this step's LSTM cell, or null
    float* xh_out;              //   + [R][2H]: the inp half of every row (encoder + C(comm) + biases, the gate product's left operand), or null
    const int32_t* alive_in;    // [R] or null (t = 0: everyone alive, quirk Q21)
    const int32_t* comm_in;     // [R] or null (gate sampled at t-1, quirk Q22)
    float* out;                 // [R][OT] log-probs | value
    int32_t* action;            // [nheads][R]
    float* obs;                 // [E][N][obs_dim] or null: next_state rows, stored from inside this kernel
    int32_t* obs_rec;           // incremental mode (ic3_env_set_incremental_obs): what the rows of `obs` hold painted, per env
    int obs_incr;               // 1: `obs` still holds exactly what obs_rec describes -> clear those entries, no zero fill
    int obs_dim;                // floats per observation row
    int ntiles;                 // workgroups = tiles: n_full tiles of EPT envs, then half tiles of EPTh envs
    int n_full, EPTh;
    // env
    int E, N, EPT, G;
    int auto_reset;             // env handle in auto-reset mode: an env with t == 0 starts an episode (h = c = 0, gate 0)
    const void* l_wp3;          // gate_split (the default): [W_ih | W_hh] as three bf16 planes in fragment order; null = fp32 instruction
    int inner;                  // comm_passes > 1 (comm.py:179): NOT the last communication pass of the step — the launch
                                // ends behind the LSTM cell (h, c updated; no heads, draws, env.step, obs rows)
    int keep_state;             // not the FIRST pass of the step: h, c hold the previous pass (an env at t == 0 keeps them)
    int tile_words;             // int32 words of one env-descriptor block in LDS
    int npass;                  // MP instantiation (ic3_policy.npasses >= 2): communication passes played inside the launch
    const ps_f32x4* c_wp_p[4];  //   packed C_modules[i].weight and encoder.bias + C_modules[i].bias of pass i
    const ps_f32x4* enc_bias_p[4];
    uint32_t seed, gid0;
    const int32_t* episode;
    const int32_t* tstep;
    StepOut so;
    PPState pp;
    TJState tj;
};

// zero-store slots of one K block: 16 slots (one behind every pair of MFMAs of a full tile), S of them in use, evenly
constexpr bool ps_zslot(int S, int i) { return S > 0 && ((i + 1) * S / 16) > (i * S / 16); }
// the same for the split-product loop (SPLIT = 1): 36 slots per 16-k block (one behind every pair of bf16 MFMAs)
constexpr bool ps_zslot36(int S, int i) { return S > 0 && ((i + 1) * S / 36) > (i * S / 36); }

// Geometry of one tile, derived from the kernel arguments and the tile index only — the phases behind the gate loop
// derive it AGAIN from a re-read copy of the arguments instead of keeping ~40 scalars (and the 40 argument pointers)
// alive across the loop: round 2's kernel spilled 150 SGPRs to VGPR lanes and read them back with ~800 v_readlane per wave.
struct TileGeom {
    int e0, nenv, rows;
    bool two, obs_here;
    size_t r0;
    long long ob0;      // first float of the tile's obs rows
    int oL, ohead, onb, mis, c_lo, c_hi, zend;
};
template <int KIND>
__device__ __forceinline__ TileGeom tile_geom(const StepArgs& a, int tile_id)
{
    TileGeom g;
    // full tiles first; the envs left over behind the last round that gives every CU the same number of them go out as
    // HALF tiles (<= 32 rows: one 32-row MFMA tile, half the matrix work) — see plan_tiles()
    const bool half = tile_id >= a.n_full;
    g.e0 = half ? a.n_full * a.EPT + (tile_id - a.n_full) * a.EPTh : tile_id * a.EPT;
    g.nenv = min(half ? a.EPTh : a.EPT, a.E - g.e0);
    g.rows = g.nenv * a.N;                                       // valid rows of this tile (<= 64; <= 32 in a half tile)
    g.two = g.rows > 32;                                         // second 32-row MFMA tile in use (workgroup-uniform)
    g.r0 = (size_t)g.e0 * a.N;
    g.obs_here = (KIND != 0) && a.obs != nullptr;
    g.ob0 = (long long)g.e0 * a.N * a.obs_dim;
    g.oL = g.rows * a.obs_dim;                                   // floats of the tile
    g.ohead = (int)((4 - (g.ob0 & 3)) & 3);
    g.onb = g.obs_here ? (g.oL - g.ohead) >> 2 : 0;              // float4s of the 16-byte aligned body
    // The body is cut into 1 KiB-aligned chunks of 64 float4s (see pp_obs_kernel); chunk c holds body indices
    // [64c - mis, 64c - mis + 64); full chunks: [c_lo, c_hi)
    g.mis = (int)(((g.ob0 + g.ohead) >> 2) & 63);
    g.c_lo = g.mis ? 1 : 0;
    g.c_hi = (g.mis + g.onb) >> 6;
    g.zend = g.obs_here ? max(0, (64 * g.c_hi - g.mis) * 16) : 0;   // bytes of the body up to the last full chunk
    if (a.obs_incr) g.zend = 0;                      // incremental rows: nothing to zero-fill here (the host
                                                     // also sets every slot count to 0)
    return g;
}

// the kernel arguments once more, by scalar loads from the kernarg segment behind an opaque pointer (the compiler cannot
// tell that they are the values it already holds, so none of the first copy has to stay alive for the second)
__device__ __forceinline__ void reload_args(StepArgs& a)
{
    typedef const __attribute__((address_space(4))) int* KWords;
    KWords kp = (KWords)__builtin_amdgcn_kernarg_segment_ptr();
    IC3_OPAQUE_SGPR(kp);
    static_assert(sizeof(StepArgs) % 4 == 0, "StepArgs is a whole number of dwords");
    int* dst = reinterpret_cast<int*>(&a);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(StepArgs) / 4); ++i) dst[i] = kp[i];
}

// MP = 1 (round 5): comm_passes > 1 (comm.py:179-218) as a loop INSIDE the launch — masks, positions and window descriptors
// are built once; every pass gathers the encoder output again (its bias differs per pass: encoder.bias + C_i.bias, and a
// kept copy would be 32 registers or 32 KB of LDS the kernel does not have), takes h from the h half of the A tile where the
// previous pass's cell epilogue left it and c from the registers it was computed in, and only the LAST pass stores h', c',
// issues the obs zero fill and runs the back half.  MP = 0 compiles to exactly the one-pass kernel.
template <int H, int KIND, int SPLIT = 0, int MP = 0>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 2 : 1) void policy_step_kernel(const StepArgs a_in)
{
    static_assert(!MP || (SPLIT && KIND != 0), "the in-launch pass loop exists for the split gate product on an env handle");
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, BM = 64, NT = 2 * H, NW = H / 32, H4 = H / 4;
    IC3_DYNAMIC_LDS(float, smem);
    float* const As = smem;                                      // [BM][LDA]: cols [0,H) enc / comm / inp, [H,2H) h / h'
    ps_f32x4* const As4 = reinterpret_cast<ps_f32x4*>(smem);
    float* const sm = As + BM * LDA;                             // [BM] m_j = alive_j * comm_action_j
    float* const sscale = sm + BM;                               // [BM] per-env 1/(n_alive-1)
    int32_t* const sact = reinterpret_cast<int32_t*>(sscale + BM);   // [BM] env action (head 0) of every row
    uint32_t* const rmask = reinterpret_cast<uint32_t*>(sact + BM);  // [BM] window cells of every row that carry a count
    uint32_t* const sfm = rmask + BM;                            // [2] (+2 pad) bit r: row r starts an episode (auto-reset)
    int32_t* const sep = reinterpret_cast<int32_t*>(sfm + 4);    // [BM] episode counter of the tile's envs (Philox key)
    int32_t* const sts = sep + BM;                               // [BM] step counter of the tile's envs
    float* const shb = reinterpret_cast<float*>(sts + BM);       // [16] head / value biases
    float* const slb = shb + 16;                                 // [4H] b_ih + b_hh
    int32_t* const tile = reinterpret_cast<int32_t*>(slb + 4 * H);   // env descriptors of the tile's envs
    constexpr int tz = 0;
    constexpr int KB = K / 8;
    constexpr int PER = H / 16;

    // ---- state that crosses the gate loop: accumulators, the old cell state, the zero-store cursor ---------------------
    ps_f32x16 acc[2][4];
    float cold[2][16];
    __amdgpu_buffer_rsrc_t zr;                                   // descriptor of the tile's obs slice (see below)
    ps_f32x4 zv = { 0.f, 0.f, 0.f, 0.f };
    IC3_OPAQUE_VGPR(zv);                                        // keep it in registers (no re-materialisation per store)
    int zlane, zso;                                              // lane offset; running byte offset of this wave's next chunk (SGPR)
    auto zero_store = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ps_u32x4, zv), zr, zlane, zso, PS_ZSTORE_AUX);
        zso += NW * 1024;
    };

    for (int pass = 0;; ++pass) {   // (one iteration unless MP)
    // =====================================================================================================================
    // FRONT: masks, descriptors, encoder, comm, C product, gate GEMM
    // =====================================================================================================================
    {
        StepArgs a_re;                                               // MP: the arguments again for every pass (kernarg segment),
        if constexpr (MP != 0) reload_args(a_re);                    // nothing of them lives across a pass
        const StepArgs& a = MP ? a_re : a_in;
        int tid_v = threadIdx.x;
        if constexpr (MP != 0) IC3_OPAQUE_VGPR(tid_v);               // (per pass: nothing derived from it is hoisted out of the pass loop)
        const int tid = tid_v;
        const bool first = !MP || pass == 0, last = !MP || pass + 1 == a.npass;
        // (the pass's C weights / bias by scalar selects: indexing the arrays with `pass` would put the argument block in scratch)
        const ps_f32x4* const enc_bias_now = !MP ? a.enc_bias : pass == 0 ? a.enc_bias_p[0] : pass == 1 ? a.enc_bias_p[1] : pass == 2 ? a.enc_bias_p[2] : a.enc_bias_p[3];
        const ps_f32x4* const c_wp_now = !MP ? a.c_wp : pass == 0 ? a.c_wp_p[0] : pass == 1 ? a.c_wp_p[1] : pass == 2 ? a.c_wp_p[2] : a.c_wp_p[3];
        IC3_PRIO_AT(1);
        // (Two experiments on the co-residency of the two workgroups of a CU were measured in round 3 and removed again:
        //  a start stagger of the second resident, and a per-CU lock that lets one gate loop run at a time — the loops
        //  did take turns, 36 us instead of 76, and the phases around them stretched by exactly what the loops gained:
        //  an fp32 MFMA stream leaves the vector ALU to nobody.  profiles/r03/gate_lock.txt, pacing_sweep.txt.)
        const int N = a.N;
        const int WW = (KIND == 0) ? 0 : (KIND == IC3_ENV_PP) ? (2 * a.pp.v + 1) * (2 * a.pp.v + 1) : (2 * a.tj.v + 1) * (2 * a.tj.v + 1);
        const int total = a.pp.Np + a.pp.nprey;
        const int nsegE = N * WW;
        const int tjw = tj_tile_words(N, WW);
        const float inv_WW = 1.0f / (float)max(WW, 1);
        const float invN = 1.0f / (float)N, inv_nsegE = 1.0f / (float)max(nsegE, 1);   // div_small(): no integer divisions
        // env descriptors of `ne` envs starting at env `eb`, into the LDS block `tl` (two phases around a barrier):
        //   PP: sr[EPT*total] | sc[EPT*total] | tab[EPT*N*WW] (int2);  TJ: EPT x TJTile
        auto desc_positions = [&](int32_t* tl, int eb, int ne) __attribute__((always_inline)) {
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
        auto desc_tab = [&](int32_t* tl, int ne) __attribute__((always_inline)) {
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
        const TileGeom g = tile_geom<KIND>(a, blockIdx.x);
        const int lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        const int e0 = g.e0, nenv = g.nenv, rows = g.rows;
        const bool two = g.two;
        const size_t r0 = g.r0;

        // ---- dense observation of the state this step acts on (the `state` the reference hands to policy_net,
        // trainer.py:49), written by the launch that consumes it.  A wave that streams fp32 MFMAs leaves the vector ALU
        // to no other wave of its SIMD (measured: tools/exp/ws_probe.hip), so the store stream shares time with the
        // matrix work from INSIDE the same instruction stream.  The rows are ~98 % zeros: the tile's contiguous slice of
        // the obs tensor is ZERO-FILLED by stores sprinkled between the MFMAs of the gate loop and the transcendentals
        // of the LSTM epilogue, and the few non-zero entries (<= 3 per window cell) are patched in at the very end,
        // after every wave has seen its zero stores complete (s_waitcnt + barrier).
        // The (at most two) ragged chunks at the ends of the body go out with lane predicates behind the cell epilogue;
        // the full ones [c_lo, c_hi) are dealt round-robin to the waves and issued by zero_store(): ONE buffer store
        // through a descriptor that ends with chunk c_hi - 1 — a slot past the wave's last chunk is dropped by the
        // hardware range check, so a slot is two instructions with no count, compare or branch, and the number of
        // stores in flight at any point of the program is a compile-time constant (exact compiler waits, see above).
        // Cache policy: non-temporal.  1.2 GB of zeros per launch flow through the 4 MB L2s next to the 0.6 MB of
        // weights every tile streams from there: with plain stores (a -DIC3_PS_PLAIN_STORES build) the kernel takes
        // 0.50 ms instead of 0.38.
        {
            const int ws = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int first = (64 * (g.c_lo + ws) - g.mis) * 16;     // this wave's first full chunk (>= 0)
            const float* obody = a.obs ? a.obs + g.ob0 + g.ohead : nullptr;    // (no obs rows: an empty descriptor, no arithmetic on null)
            zr = make_rsrc(obody, (uint32_t)((last && a.obs) ? g.zend : 0));
            zlane = lane * 16;
            zso = __builtin_amdgcn_readfirstlane(first);
        }

        // ---- S0: masks, per-env scale (comm.py:102-107,194-196; quirks Q21/Q23), entity positions ----------------
        // auto-reset: an env whose t == 0 is at the start of an episode — no alive mask yet (everyone counts as alive,
        // quirk Q21), gate 0 (no communication on the first step, quirk Q22), zero LSTM state (trainer.py:38-51)
        const bool autor = (KIND != 0) && a.auto_reset;              // workgroup-uniform
        // EVERY global load of this phase is issued before the first LDS write that needs one (round 2/3a: seven
        // load -> wait -> write sequences one after the other, 4.6 us): the h rows first (the big one), then one clamped,
        // unconditional load per thread for each small array, then the positions, and only then the writes.
        ps_f32x4 hv[8];                                              // parked in registers until S4
        unsigned long long fmask = 0;                                // rows that start an episode: zero h / c, no masks
        int32_t* sr = tile;
        int32_t* sc = tile + a.EPT * total;
        int2* ptab = reinterpret_cast<int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
        if (first) {
        {
            // rows are contiguous: float4 number idx of the tile sits at byte 16 * idx; rows >= `rows` read as zeros
            // (descriptor range check)
            const __amdgpu_buffer_rsrc_t rhh = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<void*>(a.h + r0 * H), 0, (uint32_t)rows * H * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hv[i] = __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rhh, tid * 16 + i * NT * 16, 0, 0));
            }
        }
        const int rcl = min(tid, max(rows - 1, 0)), ecl = min(tid, max(nenv - 1, 0));   // clamped row / env of this thread
        int v_alive = 1, v_comm = 1, v_tsrow = 1, v_ep = 0, v_ts = 0;
        if (a.alive_in) v_alive = a.alive_in[r0 + rcl];              // (uniform branches: a pointer is null or it is not)
        if (a.comm_in) v_comm = a.comm_in[r0 + rcl];
        if constexpr (KIND != 0) {                                   // Philox counters of the draws (S11): read here, once
            v_tsrow = a.tstep[e0 + div_small(rcl, invN)];            // (read whether or not the handle auto-resets: a
                                                                     //  branch around it would wait for it on the spot)
            v_ep = a.episode[e0 + ecl];
            v_ts = a.tstep[e0 + ecl];
        }
        const float v_hb = a.head_b[min(tid, a.OT - 1) + tz];
        const float v_lb0 = a.l_bias[tid + tz], v_lb1 = a.l_bias[tid + NT + tz];   // 4H = 2 NT
        desc_positions(tile, e0, nenv);                              // (its LDS writes wait for everything above too)
        if (tid < BM) {                                              // wave 0, all lanes
            // auto-reset: an env whose t == 0 is at the start of an episode (see above); `sact` carries the alive flags
            // to the per-env scale below (it is the draws' action buffer much later)
            const bool in = tid < rows, fr = autor && in && v_tsrow == 0;
            sm[tid] = (in && !fr) ? (float)(v_alive * v_comm) : 0.f;
            sact[tid] = (in && a.alive_in && !fr) ? v_alive : 1;
            if (tid < 16) shb[tid] = tid < a.OT ? v_hb : 0.0f;       // (a load behind the gate loop would first wait for
                                                                     //  every zero store in flight)
            if constexpr (KIND != 0) {
                rmask[tid] = (WW <= 32) ? 0u : ~0u;                  // filled next to the window descriptors (S1)
                if (tid < nenv) {
                    sep[tid] = v_ep;
                    sts[tid] = v_ts;
                }
            }
            const unsigned long long fb = __ballot(fr);              // later phases test a bit
            if (lane == 0) {
                sfm[0] = (uint32_t)fb;
                sfm[1] = (uint32_t)(fb >> 32);
            }
        }
        slb[tid] = v_lb0;
        slb[tid + NT] = v_lb1;
        if (g.obs_here && last) {   // (behind this phase's loads: their waits count these stores as younger, they do not wait for them)
#pragma unroll 1
            for (int i = 0; i < a.z0; ++i) zero_store();
        }
        __syncthreads();
        for (int el = tid; el < nenv; el += NT) {                    // per-env 1 / (n_alive - 1) (comm.py:194-196), read at S5
            int n_alive = 0;
            for (int j = 0; j < N; ++j) n_alive += sact[el * N + j];
            sscale[el] = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
        }
        if (autor)
            fmask = (unsigned long long)__builtin_amdgcn_readfirstlane(sfm[0]) |
                    ((unsigned long long)__builtin_amdgcn_readfirstlane(sfm[1]) << 32);

        // ---- S1: window descriptors --------------------------------------------------------------------------------
        if constexpr (KIND != 0) {
            desc_tab(tile, nenv);
            __syncthreads();
        }
        }   // first
        // encoder weight rows / pre-summed location rows behind buffer descriptors (32-bit gather offsets)
        const BufRows encW = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.Wt), 0,
                                                                 (uint32_t)((size_t)a.obs_dim * H * sizeof(float)), 0x00020000),
                               a.Wt != nullptr };
        const BufRows encL = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.loc_table), 0, 0x7fffffffu, 0x00020000),
                               a.loc_table != nullptr };
        // ---- S2: encoder(obs) + C.bias as a sparse gather (comm.py:51,119; pp/tj_encode_kernel) -> inp half ----------
#pragma unroll 2
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / H4, c4 = idx - row * H4;
            ps_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (row < rows) {
                const int el = div_small(row, invN), aa = row - el * N;
                if constexpr (KIND == 0) {
                    v = *reinterpret_cast<const ps_f32x4*>(a.enc_in + (r0 + row) * H + 4 * c4);
                } else if constexpr (KIND == IC3_ENV_PP) {
                    v = pp_encode_row_t(sr + el * total, sc + el * total, ptab + el * nsegE, aa, c4, H4, WW,
                                        a.pp.dim * a.pp.dim + 4, a.pp.dim, encW, enc_bias_now + tz, encL, rmask[row]);
                } else {
                    v = tj_encode_row_t(tj_tile_at(tile + el * tjw, N), a.tj, aa, c4, H4, encW, enc_bias_now + tz, encL,
                                        rmask[row]);
                }
            }
            As4[row * LDA4 + c4] = v;
        }
        // ---- S4 (the other half of the tile: no barrier in front): h -> h half ----------------------------------------
        if (first) {   // (MP, later passes: the h half holds the previous pass's h')
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / H4, c4 = idx - row * H4;
            As4[row * LDA4 + H4 + c4] = (autor && !a.keep_state && ((fmask >> row) & 1)) ? ps_f32x4{ 0.f, 0.f, 0.f, 0.f } : hv[i];
        }
        }
        __syncthreads();

        // ---- S3: the encoder output moves into the accumulators of the C product (MFMA C/D layout:
        //      col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ------------------------------------------------
        ps_f32x16 accC[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt == 1 && !two) break;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                accC[rt][reg] = As[lr * LDA + col];
            }
        }
        // the old cell state of the tile, requested HERE (a quarter of a tile's life ahead of its use): loads of a wave
        // return in order, so a load that misses to HBM in front of the gate loop's weight stream would stall that
        // stream — here the comm phase and the C product cover it — and behind the loop it would queue up behind the
        // zero stores (one counter, in order).  Rows >= `rows` read as zeros (range check).  32 registers held through
        // the loop.
        if constexpr (!SPLIT) {
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.c + r0 * H, (uint32_t)rows * H * 4u);
            const int voff = (4 * lh * H + col) * 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (rt == 1 && !two) break;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                    cold[rt][reg] = buf_load_b32(rc, voff + lc * H * 4, 0);
                }
            }
        }
        if (g.obs_here && last) {
#pragma unroll 1
            for (int i = 0; i < a.z3; ++i) zero_store();
        }
        __syncthreads();   // every wave has its share of the encoder output

        if (g.obs_here && last) {
            // (no load is waited for in the comm phase: the acknowledgements of these run under its LDS work)
#pragma unroll 1
            for (int i = 0; i < a.zf; ++i) zero_store();
        }
        if (!a.comm_zero) {   // comm_mask_zero (comm.py:40-41): C sees zeros, inp = enc + C.bias
            // ---- S5: comm_j = m_j (S_e - m_j h_j) scale_e (closed form of comm.py:181-205) -> inp half ----------------
            {
                const int c4 = tid % H4;
                for (int el = tid / H4; el < nenv; el += NT / H4) {
                    const ps_f32x4* hp = As4 + (el * N) * LDA4 + H4 + c4;
                    const float scl = sscale[el];
                    // (one component per instruction: mask_fma4 / comm_out4 in ic3_common.hpp — the packed form of this loop was
                    //  measured wrong on gfx950 now and then)
                    ps_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
                    for (int i = 0; i < N; ++i) S = mask_fma4(sm[el * N + i], hp[i * LDA4], S);
                    for (int j = 0; j < N; ++j) {
                        const float m = sm[el * N + j];
                        As4[(el * N + j) * LDA4 + c4] = comm_out4(m, S, hp[j * LDA4], scl);
                    }
                }
                for (int idx = rows * H4 + tid; idx < BM * H4; idx += NT) {
                    const int row = idx / H4, c4p = idx - row * H4;
                    As4[row * LDA4 + c4p] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                }
            }
            // B fragments of C: lane (li, lh) of wave w reads Wp[kb][32w + li][lh] -> k = 8kb + 4lh + j, j = 0..3
            constexpr int KBC = H / 8, CH = (KBC < 8) ? KBC : 8, NCH = KBC / CH;
            // weights through a buffer descriptor: lane offset in one VGPR, the k part of the address on the scalar ALU
            const __amdgpu_buffer_rsrc_t rcw = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<ps_f32x4*>(c_wp_now), 0, (uint32_t)((size_t)H * H * sizeof(float)), 0x00020000);
            const int wlane = (col * 2 + lh) * 16;
            auto cwp = [&](int k) __attribute__((always_inline)) {
                return __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rcw, wlane, k * (H * 2 * 16), 0));
            };
            ps_f32x4 cb[2][CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) cb[0][k] = cwp(k);
            __syncthreads();
            // ---- S6: accC (= enc) += comm . C.weight^T -----------------------------------------------------------------
            auto cprod = [&](auto two_c) __attribute__((always_inline)) {
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
                            // (the C product's 32 accumulators stay in VGPRs in every build: with IC3_PS_AGPR they would
                            //  come on top of the gate loop's 128 AGPRs and leave 96 VGPRs to everything else)
                            accC[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], cb[ch & 1][k][j], accC[0], 0, 0, 0);
                            if constexpr (TWO) accC[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], cb[ch & 1][k][j], accC[1], 0, 0, 0);
                        }
                        if (g.obs_here && last && kb < a.zc) zero_store();
                    }
                }
            };
            if (two) cprod(std::true_type{});
            else cprod(std::false_type{});
            __syncthreads();   // every wave has read the comm tile
        }

#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[rt][gt][i] = 0.0f;
        if constexpr (SPLIT != 0) {
            // ---- gate_split: the gate product as nine exact bf16 x bf16 products per 16 k-steps ---------------------------
            // B: pre-split weight planes Wp[plane][kb16][gate][wave][lane] (16 bytes = the 8 bf16 of k = 16 kb + 8 lh + i,
            // column 32 w + li of the gate); ONE 16-k block in registers (48 VGPRs), a plane refilled for the next block
            // right behind its last product of this one (products grouped by weight plane).  A: this wave's rows of the
            // fp32 tile, 8 values per row and block, split in registers.  The old cell state is requested in the last
            // block, into the plane registers that are no longer refilled (the block is common code without store slots).
            constexpr int KB16 = K / 16;
            const __amdgpu_buffer_rsrc_t rg3 = make_rsrc(a.l_wp3, (uint32_t)((size_t)3 * K * 4 * H * 2));
            const int g3lane = (w * 64 + lane) * 16;
            constexpr int GSTRIDE = NW * 64 * 16;
            auto wq3 = [&](int pl, int kb, int gt) __attribute__((always_inline)) {
                return __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, ((pl * KB16 + kb) * 4 + gt) * GSTRIDE, PS_WLOAD_AUX);
            };
            ps_u32x4 bq[3][4];
            // ---- S7: inp = enc + C.bias + C(comm) -> inp half --------------------------------------------------------
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (rt == 1 && !two) break;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    As[lr * LDA + col] = accC[rt][reg];
                }
            }
            if constexpr (KIND != 0 && MP == 0) {
                if (a.xh_out) {   // (uniform; ic3_env_set_record_out) the same values -> the inp half of the record's [inp | h] rows
                    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.xh_out + r0 * 2 * H, (uint32_t)rows * 2 * H * 4u);
                    const int xoff = (4 * lh * 2 * H + col) * 4;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        if (rt == 1 && !two) break;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const float xv = accC[rt][reg];           // (a copy: bit_cast of a vector element reads element 0)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, xv), rx,
                                                                  xoff + (32 * rt + (reg & 3) + 8 * (reg >> 2)) * 2 * H * 4, 0, PS_ZSTORE_AUX);
                        }
                    }
                }
            }
            __syncthreads();
            IC3_PRIO_AT(0);
            // (MP, later passes: the cell state the previous pass's epilogue stored to c_out — read past the vector L1, which
            //  may still hold the lines as the first pass loaded them)
            const __amdgpu_buffer_rsrc_t rc_old = make_rsrc((first ? a.c : a.c_out) + r0 * H, (uint32_t)rows * H * 4u);
            const int voff_old = (4 * lh * H + col) * 4;
            auto block3 = [&](auto two_c, auto s_c, auto refill_c, auto loadc_c, int kb) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
                constexpr int S = decltype(s_c)::value;
                constexpr bool REFILL = decltype(refill_c)::value;
                constexpr bool LOADC = decltype(loadc_c)::value;
                // Activation split in three stages around the products of the FIRST weight plane group: the most significant
                // plane of the 8 (4) value pairs, then that group's products pass by pass (most significant activation
                // plane first) with the next plane of two (one) pairs computed behind each pair of MFMAs — 4 vector
                // instructions per pair in the shadow of a 32-cycle MFMA instead of 72 (36) in front of the block's first.
                // The other two groups run fragment by fragment, least significant activation plane first, as before.
                constexpr int NRT = TWO ? 2 : 1;
                ps_u32x4 ap[2][3];
                ps_f32x2 xr[2][4];                                   // the raw values, then what the planes so far leave
                {
                    const ps_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh;
                    const ps_f32x4 x0 = s0[0], x1 = s0[1];
                    xr[0][0] = ps_f32x2{ x0[0], x0[1] }, xr[0][1] = ps_f32x2{ x0[2], x0[3] };
                    xr[0][2] = ps_f32x2{ x1[0], x1[1] }, xr[0][3] = ps_f32x2{ x1[2], x1[3] };
                    if constexpr (TWO) {
                        const ps_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh;
                        const ps_f32x4 y0 = s1[0], y1 = s1[1];
                        xr[1][0] = ps_f32x2{ y0[0], y0[1] }, xr[1][1] = ps_f32x2{ y0[2], y0[3] };
                        xr[1][2] = ps_f32x2{ y1[0], y1[1] }, xr[1][3] = ps_f32x2{ y1[2], y1[3] };
                    }
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ap[rt][0][q] = ps_hi_pair(xr[rt][q]);
                }
                auto products = [&](int pa, int pb, int gt) __attribute__((always_inline)) {
                    acc[0][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(ps_bf16x8, ap[0][pa]), __builtin_bit_cast(ps_bf16x8, bq[pb][gt]), acc[0][gt], 0, 0, 0);
                    if constexpr (TWO)
                        acc[1][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(ps_bf16x8, ap[1][pa]), __builtin_bit_cast(ps_bf16x8, bq[pb][gt]), acc[1][gt], 0, 0, 0);
                };
                auto slot = [&](int i) __attribute__((always_inline)) {                              // (folded after unrolling)
                    if (ps_zslot36(S, i)) {
                        __builtin_amdgcn_sched_barrier(0);
                        zero_store();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
#pragma unroll
                for (int pa = 0; pa < 3; ++pa) {                      // weight plane group 0, activation planes 0, 1, 2
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        products(pa, 0, gt);
                        slot(pa * 4 + gt);
                        if (pa < 2) {
#pragma unroll
                            for (int j = 0; j < NRT; ++j) {
                                const int rt = (gt * NRT + j) >> 2, q = (gt * NRT + j) & 3;
                                ap[rt][pa + 1][q] = ps_next_pair(xr[rt][q], ap[rt][pa][q]);
                            }
                        } else if constexpr (REFILL) {
                            bq[0][gt] = wq3(0, kb + 1, gt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    if (pb > 0) {
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt) {
                            // the six products of ONE weight fragment back to back (three activation terms, least
                            // significant first, two row tiles), then its refill: 11 fragments x 6 MFMAs of lead
#pragma unroll
                            for (int pa = 2; pa >= 0; --pa) {
                                products(pa, pb, gt);
                                slot((pb * 4 + gt) * 3 + (2 - pa));
                            }
                            if constexpr (REFILL) bq[pb][gt] = wq3(pb, kb + 1, gt);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (LOADC) {                            // 11 + 11 + 10 old cell states behind the three plane groups
#pragma unroll
                        for (int q = 0; q < 11; ++q) {
                            const int idx = 11 * pb + q;
                            if (idx < 32) {
                                const int rt = idx >> 4, reg = idx & 15;
                                const int so = (32 * rt + (reg & 3) + 8 * (reg >> 2)) * H * 4;
                                if (TWO || rt == 0) {
                                    if (first) cold[rt][reg] = buf_load_b32(rc_old, voff_old, so);
                                    else cold[rt][reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc_old, voff_old, so, 17));
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto gate_loop3 = [&](auto two_c, auto s_c) __attribute__((always_inline)) {
                // (the first block's planes are requested inside the store-slot variant, behind an opaque zero: in front of
                //  the switch the compiler copies / spills the 12 fragments into every variant's own registers)
                int zo = 0;
                IC3_OPAQUE_SGPR(zo);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt)
                        bq[pl][gt] = __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, ((pl * KB16) * 4 + gt) * GSTRIDE + zo, PS_WLOAD_AUX);
#pragma unroll 1
                for (int kb = 0; kb < KB16 - 1; ++kb) block3(two_c, s_c, std::true_type{}, std::false_type{}, kb);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto gate_loop3_s = [&](auto two_c) __attribute__((always_inline)) {
                switch ((g.obs_here && last) ? a.zs : 0) {   // workgroup-uniform; a 16-k block carries twice the slots of an 8-k block
                case 1: gate_loop3(two_c, std::integral_constant<int, 2>{}); break;
                case 2: gate_loop3(two_c, std::integral_constant<int, 4>{}); break;
                case 3: gate_loop3(two_c, std::integral_constant<int, 6>{}); break;
                case 4: gate_loop3(two_c, std::integral_constant<int, 8>{}); break;
                case 5: gate_loop3(two_c, std::integral_constant<int, 10>{}); break;
                case 6: gate_loop3(two_c, std::integral_constant<int, 12>{}); break;
                case 7: gate_loop3(two_c, std::integral_constant<int, 14>{}); break;
                case 8: gate_loop3(two_c, std::integral_constant<int, 16>{}); break;
                case 10: gate_loop3(two_c, std::integral_constant<int, 20>{}); break;
                case 12: gate_loop3(two_c, std::integral_constant<int, 24>{}); break;
                case 16: gate_loop3(two_c, std::integral_constant<int, 32>{}); break;
                default: gate_loop3(two_c, std::integral_constant<int, 0>{}); break;
                }
                // the last block: common code, no store slots, requests the old cell state
                block3(two_c, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{}, KB16 - 1);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (two) gate_loop3_s(std::true_type{});
            else gate_loop3_s(std::false_type{});
        } else {
            // gate weights in the layout Wq[k][c] = float4 (W[c][k], W[H+c][k], W[2H+c][k], W[3H+c][k]) of ic3_policy_pack:
            // ONE 16-byte load per lane feeds a k-step of all four gates.  The operand ring is PS_RING float4 deep (8 = two
            // K blocks = 32 registers, what round 2 held as two buffers of four float4 per gate); a slot is refilled right
            // behind the 8 MFMAs that read it, RING - 1 k sub-steps (56 MFMAs) ahead of its next use — with the obs zero stores
            // in flight the L2 answers slower than an idle one, a ring of 4 (24 MFMAs ahead) ran the gate loop 5 % slower.
            constexpr int RING = PS_RING;
            static_assert(RING == 4 || RING == 8, "operand ring: one or two K blocks");
            const __amdgpu_buffer_rsrc_t rgw = make_rsrc(a.l_wp, (uint32_t)((size_t)K * 4 * H * sizeof(float)));
            const int glane = (4 * lh * H + col) * 16;               // k = 8 kb + 4 lh + j (must match the A fragments)
            auto wq = [&](int kb, int j) { return buf_load_b128(rgw, glane, (8 * kb + j) * (H * 16)); };
            ps_f32x4 wk[RING];
    #pragma unroll
            for (int i = 0; i < RING; ++i) wk[i] = wq(i >> 2, i & 3);
            __builtin_amdgcn_sched_barrier(0);
            // ---- S7: inp = enc + C.bias + C(comm) -> inp half ------------------------------------------------------------
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
            IC3_PRIO_AT(0);

            // ---- S8: gates = [inp | h] . [W_ih | W_hh]^T (comm.py:215, torch.nn.LSTMCell; the bias joins in the epilogue) --
            // `SB` = first ring slot of this K block, REFILL = the ring is refilled for block kb + RING / 4.  The compiler's
            // waits in front of each k sub-step come out exact: vmcnt(RING - 1 + stores issued since the slot's refill).
            auto block = [&](auto two_c, auto s_c, auto sb_c, auto refill_c, int kb) {
                constexpr bool TWO = decltype(two_c)::value;
                constexpr int S = decltype(s_c)::value;
                constexpr int SB = decltype(sb_c)::value;
                constexpr bool REFILL = decltype(refill_c)::value;
                const ps_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
                ps_f32x4 a1;
                if constexpr (TWO) a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
    #pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        mfma_acc(acc[0][gt], a0[j], wk[SB + j][gt]);
                        if constexpr (TWO) mfma_acc(acc[1][gt], a1[j], wk[SB + j][gt]);
                        // a store slot is two instructions that wait for nothing: it rides in the 64-cycle shadow of an MFMA
                        if (ps_zslot(S, 4 * j + gt)) {                // (folded after unrolling)
                            __builtin_amdgcn_sched_barrier(0);        // pinned between the MFMAs it follows / precedes
                            zero_store();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (REFILL) wk[SB + j] = wq(kb + RING / 4, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto gate_loop = [&](auto two_c, auto s_c) {
                static_assert(KB % 2 == 0 && KB >= 4, "K/8 must be even");
                constexpr std::integral_constant<int, 0> s0{};
                constexpr std::integral_constant<int, RING == 8 ? 4 : 0> s1{};
    #pragma unroll 1
                for (int kb = 0; kb < KB - 2; kb += 2) {
                    block(two_c, s_c, s0, std::true_type{}, kb);
                    block(two_c, s_c, s1, std::true_type{}, kb + 1);
                }
                block(two_c, s_c, s0, std::integral_constant<bool, RING == 4>{}, KB - 2);
                block(two_c, s_c, s1, std::false_type{}, KB - 1);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto gate_loop_s = [&](auto two_c) {
                switch (g.obs_here ? a.zs : 0) {   // workgroup-uniform
                case 1: gate_loop(two_c, std::integral_constant<int, 1>{}); break;
                case 2: gate_loop(two_c, std::integral_constant<int, 2>{}); break;
                case 3: gate_loop(two_c, std::integral_constant<int, 3>{}); break;
                case 4: gate_loop(two_c, std::integral_constant<int, 4>{}); break;
                case 5: gate_loop(two_c, std::integral_constant<int, 5>{}); break;
                case 6: gate_loop(two_c, std::integral_constant<int, 6>{}); break;
                case 7: gate_loop(two_c, std::integral_constant<int, 7>{}); break;
                case 8: gate_loop(two_c, std::integral_constant<int, 8>{}); break;
                case 10: gate_loop(two_c, std::integral_constant<int, 10>{}); break;
                case 12: gate_loop(two_c, std::integral_constant<int, 12>{}); break;
                case 16: gate_loop(two_c, std::integral_constant<int, 16>{}); break;
                default: gate_loop(two_c, std::integral_constant<int, 0>{}); break;
                }
            };
            if (two) gate_loop_s(std::true_type{});
            else gate_loop_s(std::false_type{});
        }
        IC3_PRIO_AT(2);
    }

    // =====================================================================================================================
    // BACK: LSTM cell epilogue, heads, draws, env.step, obs patches.  The kernel arguments are read AGAIN from the kernarg
    // segment (scalar loads behind an opaque pointer) and everything derived from them or from the thread index is derived
    // again, so that nothing of it occupies registers across the gate loop.
    // =====================================================================================================================
    {
        StepArgs a;
        reload_args(a);
        int tid = threadIdx.x;
        IC3_OPAQUE_VGPR(tid);
        const bool first = !MP || pass == 0, last = !MP || pass + 1 == a.npass;
        const TileGeom g = tile_geom<KIND>(a, blockIdx.x);
        const int lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        const int e0 = g.e0, nenv = g.nenv, rows = g.rows;
        const bool two = g.two, obs_here = g.obs_here && last;
        const size_t r0 = g.r0;
        const int N = a.N;
        const int WW = (KIND == 0) ? 0 : (KIND == IC3_ENV_PP) ? (2 * a.pp.v + 1) * (2 * a.pp.v + 1) : (2 * a.tj.v + 1) * (2 * a.tj.v + 1);
        const int total = a.pp.Np + a.pp.nprey;
        const int nsegE = N * WW;
        const int tjw = tj_tile_words(N, WW);
        const float invN = 1.0f / (float)N;
        const bool autor = (KIND != 0) && a.auto_reset;
        (void)li;
        (void)total;
        (void)tjw;

        // ---- S9: LSTM cell epilogue (gate order i,f,g,o); c', h' to HBM, h' also into the h half for the heads --------
        {
            // c / h rows of the tile through buffer descriptors: one 32-bit lane offset + a constant per element instead
            // of a 64-bit address pair each, and the hardware range check (num_records = the tile's valid rows) stands
            // in for the `row < rows` predicates — an out-of-range store is dropped.
            // (MP, an inner pass: h' stays in the A tile, its stores are dropped by an empty range; c' goes to c_out — the next
            //  pass reads it back in its last gate block — instead of occupying 32 registers across that pass)
            const uint32_t nrec = (uint32_t)rows * H * 4u;
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.c_out + r0 * H), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.h_out + r0 * H), 0, last ? nrec : 0u, 0x00020000);
            const int voff = (4 * lh * H + col) * 4;
            const float bi = slb[col], bf = slb[H + col], bg = slb[2 * H + col], bo = slb[3 * H + col];
            // (cold[]: requested in front of the C product; every load this wave issued after them has been waited for
            // in the gate loop and loads return in order, so they have landed)
            if (autor && !a.keep_state && first) {
                const unsigned long long fmask = (unsigned long long)__builtin_amdgcn_readfirstlane(sfm[0]) |
                                                 ((unsigned long long)__builtin_amdgcn_readfirstlane(sfm[1]) << 32);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if ((fmask >> (32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh)) & 1) cold[rt][reg] = 0.0f;
            }
            __syncthreads();   // every wave is done with the A tile
            // head / value weights -> rows [0, OT) of the inp half: requested now, written to LDS behind the element loop
            // (the compiler's wait there allows the >= 32 stores issued meanwhile to stay in flight)
            const __amdgpu_buffer_rsrc_t rhw = make_rsrc(a.head_w, (uint32_t)(a.OT * H * sizeof(float)));
            const ps_f32x4 hw0 = buf_load_b128(rhw, tid * 16, 0), hw1 = buf_load_b128(rhw, (tid + NT) * 16, 0);
            static_assert(16 * H4 <= 2 * NT, "head weights: two float4 per thread");
            // The element loop, one copy per number of zero-store slots per element (workgroup-uniform): with the slot count
            // a compile-time constant the 16 elements of a row tile are ONE basic block, and the scheduler overlaps the
            // transcendental chains (exp -> rcp -> exp -> rcp) of neighbouring elements instead of running them end to end.
            // (GS: ic3_env_set_gates_out — the activated gates go to the update half's record, so its backward reads them
            //  instead of running the gate product again; stores only where armed, 4 x 4 bytes per element)
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<void*>(a.gates_out ? a.gates_out + r0 * 4 * H : a.c_out), 0, a.gates_out ? 4u * nrec : 0u, 0x00020000);
            const int goff = (4 * lh * 4 * H + col) * 4;
            auto cell = [&](auto ze_c, auto gs_c) __attribute__((always_inline)) {
                constexpr int ZE = decltype(ze_c)::value;
                constexpr bool GS = decltype(gs_c)::value;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if (rt == 1 && !two) break;          // half tile: rows 32..63 are padding (their h' is never read)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        const float gi = acc[rt][0][reg] + bi, gf = acc[rt][1][reg] + bf;
                        const float gg = acc[rt][2][reg] + bg, go = acc[rt][3][reg] + bo;
                        // (explicit fma: the three copies of this loop must round alike — a*b + c*d left to the compiler
                        //  comes out as fma(a, b, c*d), fma(c, d, a*b) or two products and a sum depending on the schedule)
                        const float si = fast_sigmoid(gi), tg = fast_tanh(gg), sf = fast_sigmoid(gf), so = fast_sigmoid(go);
                        const float ig = si * tg;
                        const float c1 = __builtin_fmaf(sf, cold[rt][reg], ig);
                        const float h1 = so * fast_tanh(c1);
                        // what the gate loop left of the zero fill goes out between the transcendental work of the cell
                        // (<= 2 x 32 slots, then the rest)
                        if constexpr (ZE > 0) zero_store();
                        if constexpr (ZE > 1) zero_store();
                        const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c1), rc, voff + lc * H * 4, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, h1), rh, voff + lc * H * 4, 0, 0);
                        if constexpr (GS) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, si), rg, goff + lc * 4 * H * 4, 0, PS_ZSTORE_AUX);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sf), rg, goff + lc * 4 * H * 4 + H * 4, 0, PS_ZSTORE_AUX);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, tg), rg, goff + lc * 4 * H * 4 + 2 * H * 4, 0, PS_ZSTORE_AUX);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, so), rg, goff + lc * 4 * H * 4 + 3 * H * 4, 0, PS_ZSTORE_AUX);
                        }
                        As[lr * LDA + H + col] = h1;
                    }
                }
            };
            const int ze = obs_here ? a.zepi : 0;
            // (the host arms gates_out for the SPLIT, one-pass, env instantiations only, and leaves the epilogue no zero-store
            //  slots then)
            constexpr bool GS_BUILT = SPLIT == 1 && MP == 0 && KIND != 0;
            if (GS_BUILT && a.gates_out) {
                if constexpr (GS_BUILT) cell(std::integral_constant<int, 0>{}, std::true_type{});
            } else if (ze <= 0) cell(std::integral_constant<int, 0>{}, std::false_type{});
            else if (ze == 1) cell(std::integral_constant<int, 1>{}, std::false_type{});
            else cell(std::integral_constant<int, 2>{}, std::false_type{});
            if (tid < a.OT * H4) As4[(tid / H4) * LDA4 + tid % H4] = hw0;
            if (tid + NT < a.OT * H4) As4[((tid + NT) / H4) * LDA4 + (tid + NT) % H4] = hw1;
            if (obs_here && !a.obs_incr) {
#pragma unroll 1
                for (int i = 0; i < a.zrest; ++i) zero_store();     // obs-dominated shapes
                // ragged chunks: chunk 0 when the body starts inside it, chunk c_hi when the body ends inside it; the
                // <= 3 floats in front of / behind the 16-byte aligned body
                ps_f32x4* const obody = reinterpret_cast<ps_f32x4*>(a.obs + g.ob0 + g.ohead);
                const int ws = tid >> 6;
                const int q0 = lane - g.mis, q1 = 64 * g.c_hi - g.mis + lane;
                if (ws == 0 && g.mis && q0 >= 0 && q0 < g.onb) obody[q0] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                if (ws == 1 % NW && ((g.mis + g.onb) & 63) && (g.c_hi > 0 || !g.mis) && q1 >= 0 && q1 < g.onb)
                    obody[q1] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                const int otail = (g.oL - g.ohead) & 3;
                if (tid < g.ohead) a.obs[g.ob0 + tid] = 0.f;
                if (tid < otail) a.obs[g.ob0 + g.ohead + 4 * (long long)g.onb + tid] = 0.f;
            }
        }
        __syncthreads();
        IC3_PRIO_AT(4);
        if (MP && !last) continue;          // (uniform) the next communication pass of the step
        if (a.inner) return;   // (uniform)

        // ---- S10: heads + value head (comm.py:228,239) as a 64 x 16 x H product on v_mfma_f32_16x16x4_f32: row tile of
        //      16 rows per wave, the OT <= 16 output columns are the weight rows [0, 16) of the inp half (rows >= OT hold
        //      stale finite data and only feed output columns nobody reads).  Operand layout of the instruction: A[i][k]
        //      in lane 16k + i, B[k][j] in lane 16k + j, D[4(l/16) + v][l % 16] in element v of lane l; one ds_read_b128
        //      per operand feeds four k-steps (k = 16 sg + 4 (l/16) + j — any k order is valid as long as A and B agree).
        // logits of row r -> rows [16, ..) of the inp half: z(r, o) = As[(16 + r / PER) * LDA + (r % PER) * 16 + o]
        if (obs_here) {
#pragma unroll 1
            for (int i = 0; i < a.zh; ++i) zero_store();         // (LDS + matrix work only in this phase: nothing waits for them)
        }
        {
            const int l16 = lane & 15, kq = lane >> 4;
            const float hb = shb[l16];
            for (int rtile = w; rtile < BM / 16; rtile += NW) {
                if (16 * rtile >= rows) break;
                ps_f32x4 z = { hb, hb, hb, hb };
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
                        As[(16 + r / PER) * LDA + (r % PER) * 16 + l16] = z[v];
                    }
                }
            }
        }
        __syncthreads();

        // ---- S11: log_softmax per head + the action draws (action_utils.py:32-36; same arithmetic and Philox counters
        //      as lstm_cell_heads_kernel / sample_actions_env_kernel), one task per (row, head) + one per row for the value
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
                const uint32_t x = philox_x24(a.seed, a.gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)sep[el],
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
                a.action[(size_t)hd * R + grow] = act;
                if (hd == 0) sact[tr] = act;
            }
        }
        __syncthreads();

        // ---- S12: env.step for the tile's envs with the env-action head (env_wrappers.py:76-77) ------------------------
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
        if (obs_here) {
            // every zero store of this workgroup has completed (own stores: vmcnt(0); the others': barrier) before the
            // first non-zero entry goes out to the same lines
            IC3_WAIT_VMEM();
            __syncthreads();
            float* orow0 = a.obs + g.ob0;
            // Incremental rows (opt-in experiment, ic3_env_set_incremental_obs): the caller's buffer still holds what the
            // previous call painted; its descriptors were recorded per env (PP: the window table, TJ: alive flags + window
            // table).  Clear exactly those entries, then paint — instead of zero-filling 145 KB per env for ~270 entries.
            if constexpr (KIND == IC3_ENV_PP) {
                const int2* ptab = reinterpret_cast<const int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
                const int vocab = a.pp.dim * a.pp.dim + 4;
                const size_t rec0 = (size_t)e0 * nsegE;              // the tile's first record (a.obs_rec may be null)
                if (a.obs_incr) {
                    for (int sg = tid; sg < nenv * nsegE; sg += NT) {
                        const int2 d = reinterpret_cast<const int2*>(a.obs_rec)[rec0 + sg];
                        float* cell = orow0 + (size_t)sg * vocab;
                        cell[d.x] = 0.f;
                        if (d.x != vocab - 2 && (d.y >> 16) != 0) cell[vocab - 2] = 0.f;
                        if (d.x != vocab - 1 && (d.y & 0xffff) != 0) cell[vocab - 1] = 0.f;
                    }
                    IC3_WAIT_VMEM();   // a cleared word may be painted again by another thread
                    __syncthreads();
                }
                for (int sg = tid; sg < nenv * nsegE; sg += NT) {   // descriptors of the INPUT state (S1)
                    const int2 d = ptab[sg];
                    pp_obs_patch(orow0 + (size_t)sg * vocab, d, vocab);   // (env_device.hpp)
                    if (a.obs_rec) reinterpret_cast<int2*>(a.obs_rec)[rec0 + sg] = d;
                }
            } else if constexpr (KIND == IC3_ENV_TJ) {
                const int obs_dim = a.obs_dim;
                const int recw = N + 2 * nsegE;                      // per env: alive[N] | tab[N * WW] (int2)
                if (a.obs_incr) {
                    for (int sg = tid; sg < nenv * (nsegE + N); sg += NT) {
                        const int el = div_small(sg, 1.0f / (float)(nsegE + N)), q = sg - el * (nsegE + N);
                        const int32_t* r = a.obs_rec + (size_t)(e0 + el) * recw;
                        float* env_rows = orow0 + (size_t)el * N * obs_dim;
                        if (q < N) {
                            if (!r[q]) continue;
                            float* row = env_rows + (size_t)q * obs_dim;
                            row[0] = 0.f;
                            row[1] = 0.f;
                            if (a.tj.hdr == 4) {
                                row[2] = 0.f;
                                row[3] = 0.f;
                            }
                        } else {
                            const int qq = q - N, car = div_small(qq, 1.0f / (float)WW), cellx = qq - car * WW;
                            if (!r[car]) continue;
                            const int2 d = reinterpret_cast<const int2*>(r + N)[qq];
                            float* cell = env_rows + (size_t)car * obs_dim + a.tj.hdr + (size_t)cellx * a.tj.vocab;
                            if (d.x >= 0) cell[d.x] = 0.f;
                            if (d.x != a.tj.car_class && d.y != 0) cell[a.tj.car_class] = 0.f;
                        }
                    }
                    IC3_WAIT_VMEM();
                    __syncthreads();
                }
                for (int sg = tid; sg < nenv * (nsegE + N); sg += NT) {
                    const int el = div_small(sg, 1.0f / (float)(nsegE + N)), q = sg - el * (nsegE + N);
                    const TJTile t = tj_tile_at(tile + el * tjw, N);
                    tj_obs_patch(t, a.tj, orow0 + (size_t)el * N * obs_dim, obs_dim, WW, q);
                    if (a.obs_rec) {
                        int32_t* r = a.obs_rec + (size_t)(e0 + el) * recw;
                        if (q < N) r[q] = t.sal[q];
                        else reinterpret_cast<int2*>(r + N)[q - N] = t.tab[q - N];
                    }
                }
            }
        }
    }
    break;
    }   // pass
}


static int device_cus()
{
    static int cus[64] = { 0 };   // per device (a process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

// Tile plan (a.E, a.N, a.EPT set).  Two workgroups share a CU and all tiles cost the same, so a launch whose tile count
// is not a multiple of the slot count ends with a round in which some CUs still hold two tiles while others hold one
// or none (PP-hard: 1366 tiles = 2.67 rounds of 512 slots cost 3).  Plan B: as many FULL tiles (EPT envs, two 32-row
// MFMA tiles) as give every CU the same number, the rest as HALF tiles (EPTh = floor(32 / N) envs, one MFMA tile) that
// are dispatched last and land next to a CU's last full tile (or alone).  A half tile is not half the time — the phases
// around the MFMA loops and the weight stream stay — so plan B is chosen only when a cost model says it ends earlier.
// The model's five numbers (what a CU takes for a pair of full tiles = 1, a full + a half, a pair of halves, a lone
// full, a lone half) are MEASURED on the device the first time a (device, hid_size, agents) shape is planned:
// calibrate_tile_costs() times the policy half of the kernel (KIND 0, scratch operands) on launches that put exactly that
// mix on every CU.  Round 2 shipped constants fitted by hand on one box (0.91 / 0.70 / 0.6 / 0.47); they remain the
// fallback while a stream is being captured.  IC3_PS_HALF=0 / 1 forces plan A / B (same results either way).
struct TileCosts {
    double full_half = 0.91, half_pair = 0.70, lone_full = 0.6, lone_half = 0.47;   // relative to a pair of full tiles
    bool measured = false;
};

static double tiles_cost(const TileCosts& tc, int k_full, int k_half)
{
    double c = (k_full / 2) * 1.0;
    if (k_full & 1) {
        if (k_half > 0) {
            c += tc.full_half;
            --k_half;
        } else {
            c += tc.lone_full;
        }
    }
    return c + (k_half / 2) * tc.half_pair + (k_half & 1) * tc.lone_half;
}

template <int H, int KIND, int SPLIT = 0, int MP = 0>
static int launch_step(const StepArgs& a, int tiles, size_t lds, hipStream_t s, hipEvent_t ev0 = nullptr,
                       hipEvent_t ev1 = nullptr);
// words of the small LDS arrays behind the A tile: sm, sscale, sact, rmask [64 each], sfm [4], sep, sts [64 each], shb [16], slb [4H]
static size_t ps_lds_small(int H) { return 6 * 64 + 4 + 16 + 4 * (size_t)H; }

static int check_policy_struct(const ic3_policy* p, const char* who)
{
    if (p && p->struct_size != sizeof(ic3_policy))
        return fail(-22, std::string(who) + ": ic3_policy.struct_size is " + std::to_string(p->struct_size) + ", this library's is " +
                             std::to_string(sizeof(ic3_policy)) + " (built against another ic3_rollout.h?)");
    return 0;
}

static int fill_policy(StepArgs& a, const ic3_policy* p, const char* who)
{
    if (int rc = check_policy_struct(p, who)) return rc;
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
    a.inner = p->inner_pass != 0;
    a.keep_state = p->pass_index > 0;
    a.l_wp3 = (p->gate_split && p->lstm_wp3) ? p->lstm_wp3 : nullptr;
    a.npass = 1;
    if (p->npasses >= 2) {                       // every communication pass inside one launch (ic3_policy_step only)
        if (p->npasses > 4 || p->pass_index || p->inner_pass)
            return fail(-38, std::string(who) + ": npasses takes 2..4 passes in one launch, with pass_index = inner_pass = 0");
        a.npass = p->npasses;
        for (int i = 0; i < p->npasses; ++i) {
            if (!p->c_wp_pass[i] || !p->enc_bias_pass[i]) return fail(-22, std::string(who) + ": npasses without the passes' c_wp_pass / enc_bias_pass");
            a.c_wp_p[i] = reinterpret_cast<const ps_f32x4*>(p->c_wp_pass[i]);
            a.enc_bias_p[i] = reinterpret_cast<const ps_f32x4*>(p->enc_bias_pass[i]);
        }
    }
    return 0;
}


// Times `n_full` full + `n_half` half tiles of the policy half (KIND 0) on scratch operands; returns ms (< 0 on error)
static double time_tile_mix(const ic3_policy* p, int N, int n_full, int n_half, float* scratch, hipStream_t s,
                            hipEvent_t e0, hipEvent_t e1)
{
    const int H = p->H;
    StepArgs a{};
    if (fill_policy(a, p, "calibrate_tile_costs")) return -1.0;
    a.N = N;
    a.EPT = 64 / N;
    a.EPTh = 32 / N;
    a.n_full = n_full;
    a.ntiles = n_full + n_half;
    a.E = n_full * a.EPT + n_half * a.EPTh;
    a.G = 1;
    const size_t R = (size_t)a.E * N;
    a.enc_in = scratch;
    a.h = a.h_out = scratch + R * H;
    a.c = a.c_out = scratch + 2 * R * H;
    a.out = scratch + 3 * R * H;
    const size_t lds = ((size_t)64 * (2 * H + 4) + ps_lds_small(H)) * sizeof(float);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        if (hipEventRecord(e0, s) != hipSuccess) return -1.0;
        const int rc = H == 128 ? launch_step<128, 0>(a, a.ntiles, lds, s) : launch_step<64, 0>(a, a.ntiles, lds, s);
        if (rc || hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0;
        if (rep > 0 && ms < best) best = ms;      // (the first launch warms caches / clocks)
    }
    return best;
}

static void calibrate_tile_costs(const ic3_policy* p, int N, int cus, hipStream_t s, TileCosts& tc)
{
    const int H = p->H, EPT = 64 / N, EPTh = 32 / N;
    const size_t Rmax = (size_t)2 * cus * EPT * N;
    float* scratch = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc(&scratch, (Rmax * (3 * H + 16)) * sizeof(float)) != hipSuccess) return;
    bool ok = hipMemsetAsync(scratch, 0, (Rmax * (3 * H + 16)) * sizeof(float), s) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        const double pair = time_tile_mix(p, N, 2 * cus, 0, scratch, s, e0, e1);
        const double fh = time_tile_mix(p, N, cus, cus, scratch, s, e0, e1);
        const double hh = time_tile_mix(p, N, 0, 2 * cus, scratch, s, e0, e1);
        const double lf = time_tile_mix(p, N, cus, 0, scratch, s, e0, e1);
        const double lh = time_tile_mix(p, N, 0, cus, scratch, s, e0, e1);
        if (pair > 0 && fh > 0 && hh > 0 && lf > 0 && lh > 0) {
            tc.full_half = fh / pair;
            tc.half_pair = hh / pair;
            tc.lone_full = lf / pair;
            tc.lone_half = lh / pair;
            tc.measured = true;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(scratch);
    (void)EPTh;
}

static const TileCosts& tile_costs(const ic3_policy* p, int N, int H, hipStream_t s)
{
    struct Key {
        int dev, H, N;
        TileCosts tc;
    };
    static std::vector<Key> cache;
    static const TileCosts fallback{};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fallback;
    for (const Key& k : cache)
        if (k.dev == dev && k.H == H && k.N == N) return k.tc;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!p || H > 128 || hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return fallback;
    Key k{ dev, H, N, TileCosts{} };
    calibrate_tile_costs(p, N, device_cus(), s, k.tc);
    cache.push_back(k);
    return cache.back().tc;
}

static int plan_tiles(StepArgs& a, int H, const ic3_policy* p, hipStream_t s)
{
    static const int force = getenv("IC3_PS_HALF") ? atoi(getenv("IC3_PS_HALF")) : -1;
    const int cus = device_cus();
    const int n_all = (a.E + a.EPT - 1) / a.EPT;
    a.EPTh = 32 / a.N;
    a.n_full = n_all;
    a.ntiles = n_all;
    if (a.EPTh < 1 || H > 128 || force == 0) return a.ntiles;
    const int n_full = (a.E / a.EPT) / cus * cus;               // every CU the same number of full tiles
    const int rem = a.E - n_full * a.EPT;
    const int n_half = (rem + a.EPTh - 1) / a.EPTh;
    bool plan_b = force == 1;
    if (force < 0) {
        const TileCosts& tc = tile_costs(p, a.N, H, s);
        plan_b = tiles_cost(tc, n_full / cus, (n_half + cus - 1) / cus) < tiles_cost(tc, (n_all + cus - 1) / cus, 0) - 1e-9;
    }
    if (plan_b) {
        a.n_full = n_full;
        a.ntiles = n_full + n_half;
    }
    return a.ntiles;
}

template <int H, int KIND, int SPLIT, int MP>
static int launch_step(const StepArgs& a, int tiles, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(&policy_step_kernel<H, KIND, SPLIT, MP>), lds));   // per (kernel, device)
    // one workgroup per tile, dispatched in tile order (full tiles first, see plan_tiles): the hardware dispatcher
    // balances them over the CUs (a fixed resident set walking a strided tile list was measured slower)
    const int grid = tiles;
    if (ev0 || ev1) {   // timed launch: the dispatch itself stamps the events (no separate record packets around it)
        hipExtLaunchKernelGGL((policy_step_kernel<H, KIND, SPLIT, MP>), dim3(grid), dim3(2 * H), lds, s, ev0, ev1, 0, a);
    } else {
        hipLaunchKernelGGL((policy_step_kernel<H, KIND, SPLIT, MP>), dim3(grid), dim3(2 * H), lds, s, a);
    }
    IC3_HIP(hipGetLastError());
    return 0;
}

}  // namespace ic3

using namespace ic3;


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
    const size_t lds = ((size_t)64 * (2 * H + 4) + ps_lds_small(H) + tile_words) * sizeof(float);
    const size_t limit = (H <= 128) ? 80 * 1024 : 160 * 1024;   // two workgroups per CU up to H = 128
    return lds <= limit ? (int)lds : 0;
}

extern "C" int ic3_policy_step_supported(const ic3_env* env, int H) { return policy_step_lds(env, H, 0, nullptr); }


extern "C" int ic3_policy_forward(const ic3_policy* p, const float* enc, int E, int N, float* h, float* c,
                                  const int32_t* alive_in, const int32_t* comm_in, float* out, ic3_stream stream)
{
    ic3::Range range_("ic3_policy_forward");
    if (int src = check_policy_struct(p, "ic3_policy_forward")) return src;   // before any other field is read
    if (!p || !enc || !h || !c || (!out && !p->inner_pass) || E <= 0 || N <= 0)
        return fail(-22, "ic3_policy_forward: bad arguments");
    const int H = p->H;
    if ((H != 64 && H != 128 && H != 256) || N > 64)
        return fail(-38, "ic3_policy_forward: needs hid_size 64/128/256 and <= 64 agents per env");
    StepArgs a{};
    int rc = fill_policy(a, p, "ic3_policy_forward");
    if (rc) return rc;
    if (a.npass > 1) return fail(-38, "ic3_policy_forward: npasses >= 2 is ic3_policy_step's (one call per pass here)");
    a.enc_in = enc;
    a.h = a.h_out = h;
    a.c = a.c_out = c;
    a.alive_in = alive_in;
    a.comm_in = comm_in;
    a.out = out;
    a.E = E;
    a.N = N;
    a.EPT = 64 / N;
    a.G = 1;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = plan_tiles(a, H, p, s);
    const size_t lds = ((size_t)64 * (2 * H + 4) + ps_lds_small(H)) * sizeof(float);
    if (a.l_wp3) {   // the gate product as exact bf16 split products (ic3_policy.gate_split), as in ic3_policy_step
        if (H == 128) return launch_step<128, 0, 1>(a, tiles, lds, s);
        if (H == 64) return launch_step<64, 0, 1>(a, tiles, lds, s);
        return launch_step<256, 0, 1>(a, tiles, lds, s);
    }
    if (H == 128) return launch_step<128, 0>(a, tiles, lds, s);
    if (H == 64) return launch_step<64, 0>(a, tiles, lds, s);
    return launch_step<256, 0>(a, tiles, lds, s);
}

extern "C" int ic3_policy_step(ic3_env* env, const ic3_policy* p, float* h, float* c, const int32_t* alive_in,
                               const int32_t* comm_in, float* out, int32_t* action, float* obs, float* reward,
                               int32_t* done, int32_t* alive, int32_t* is_completed, ic3_stream stream)
{
    ic3::Range range_("ic3_policy_step");
    if (int src = check_policy_struct(p, "ic3_policy_step")) return src;      // before any other field is read
    if (!env || !p || !h || !c) return fail(-22, "ic3_policy_step: null argument");
    const bool inner = p->inner_pass != 0;                       // a non-final communication pass: h, c only
    // one-shot outputs armed on the handle (ic3_env_set_hidden_out): consumed by this call whatever becomes of it
    float* const armed_h = env->h_out;
    float* const armed_c = env->c_out;
    float* const armed_g = env->gates_out;                       // (ic3_env_set_record_out: one-shot as well)
    float* const armed_x = env->xh_out;
    if (!inner) env->h_out = env->c_out = env->gates_out = env->xh_out = nullptr;
    if (!inner && (!out || !action || !reward || !done)) return fail(-22, "ic3_policy_step: null argument");
    if (inner) obs = nullptr;
    if (env->resets == 0) return fail(-22, "ic3_policy_step: reset() has not been called");
    if (!p->enc_wt || !p->enc_bias) return fail(-22, "ic3_policy_step: incomplete ic3_policy (encoder)");
    const int H = p->H;
    // next_state rows are stored from inside the kernel when their descriptors fit in LDS next to the tile's own (otherwise
    // ic3_env_observe runs as a launch of its own in front of the kernel: same rows)
    int tile_words = 0;
    int lds = obs ? policy_step_lds(env, H, 1, &tile_words) : 0;
    const bool fused_obs = lds != 0;
    if (!lds) lds = policy_step_lds(env, H, 0, &tile_words);
    if (!lds)
        return fail(-38, "ic3_policy_step: needs hid_size 64/128/256, <= 64 agents per env and an env tile that fits "
                         "in LDS (use ic3_env_encode + ic3_comm_masked_mean + GEMMs + ic3_lstm_cell_heads + ic3_env_step)");
    StepArgs a{};
    int frc = fill_policy(a, p, "ic3_policy_step");
    if (frc) return frc;
    a.h = a.h_out = h;
    a.c = a.c_out = c;
    if (!inner && armed_h && armed_c) {                          // (ic3_env_set_hidden_out): the step's LAST launch
        a.h_out = armed_h;
        a.c_out = armed_c;
    }
    if (!inner && armed_g) {
        if (!a.l_wp3 || a.npass > 1 || (H != 64 && H != 128))
            return fail(-38, "ic3_env_set_record_out: the gate record needs gate_split, one communication pass, hid_size 64 / 128");
        a.gates_out = armed_g;
        a.xh_out = armed_x;
    }
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
    const int tiles = plan_tiles(a, H, p, (hipStream_t)stream);
    // incremental obs rows (opt-in): the buffer must be the one the previous call painted, untouched since
    const bool incr = fused_obs && env->obs_rec != nullptr;
    const bool incr_valid = incr && env->painted_valid && env->painted_obs == obs;   // (no zero fill in the launch)
    a.obs_rec = incr ? env->obs_rec : nullptr;
    a.obs_incr = incr_valid ? 1 : 0;
    if (incr) {
        env->painted_obs = obs;
        env->painted_valid = true;
    }
    a.tile_words = tile_words;
    a.obs = fused_obs ? obs : nullptr;
    a.obs_dim = env->dims.obs_dim;
    a.auto_reset = env->auto_max_steps > 0;
    {   // pacing of the obs zero fill (speed only: a slot past a wave's last chunk is dropped by the hardware).
        // A wave of a full tile owns `per_wave` 1 KiB chunks.  Stores issued back to back are exposed at the HBM write
        // rate; stores between MFMAs ride in their shadows until the rate all CUs ask for exceeds what HBM takes.
        // The split below (stores per K block of the gate loop; inside the C product; in front of the comm phase; per element
        // of the cell epilogue) is the outcome of the sweeps of rounds 3-4 (profiles/r03/pacing_sweep.txt, profiles/r04).
        constexpr int zfrac = 70;                                // share of a large slice that goes out inside the gate loop (%)
        const int NWv = H / 32, KBv = 2 * H / 8;
        const long long chunks = ((long long)a.EPT * a.N * a.obs_dim / 4 + 63) / 64 + 1;   // 1 KiB chunks of a full tile
        const long long per_wave = (chunks + NWv - 1) / NWv;
        static const int ZS_SET[] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16 };
        // Small obs slices (TJ: <= 128 chunks per wave, a few per K block) go out entirely inside the gate loop — measured
        // on TJ-hard: 3 per block and nothing elsewhere 0.564 ms, the split below 0.578, none in the loop 0.605.  Large
        // ones (PP-hard: 228 per wave) would ask HBM for more than it takes while every CU is inside its gate loop:
        // 70 % in the loop, 7 % each inside the C product and in front of the comm phase, the rest between the
        // transcendentals of the cell epilogue (profiles/r03/pacing_sweep.txt).
        const bool small = per_wave <= 4 * KBv;
        int zs = 0;
        if (fused_obs && !incr_valid) {
            const double want = small ? (double)((per_wave + KBv - 1) / KBv) : (double)per_wave * zfrac / 100.0 / KBv;
            double best = 1e30;
            for (int cand : ZS_SET) {
                if (cand == 1) continue;                           // (2 slots per unrolled pair of K blocks: the compiler's
                                                                   //  vmcnt comes out one short of exact for that variant)
                const double d = want > cand ? want - cand : cand - want;
                if (d < best) {
                    best = d;
                    zs = cand;
                }
            }
        }
        a.zs = zs;
        long long left = per_wave - (long long)zs * (KBv - (a.l_wp3 ? 2 : 0));   // (split loop: no slots in its last 16-k block)
        auto take = [&](int want) {
            const int n = (fused_obs && !incr_valid) ? (int)std::min<long long>(std::max<long long>(left, 0), std::max(want, 0)) : 0;
            left -= n;
            return n;
        };
        a.z0 = take(0);
        a.z3 = take(0);
        const int share = small ? 0 : (int)((per_wave * 7 + 50) / 100);
        a.zc = take(std::min(share, H / 8));
        a.zf = take(share);
        a.zh = take(0);
        a.zepi = (!fused_obs || incr_valid || a.gates_out) ? 0 : left > 32 ? 2 : left > 0 ? 1 : 0;
        left -= 32LL * a.zepi;
        a.zrest = (fused_obs && !incr_valid) ? (int)(left > 0 ? left + 1 : 0) : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (obs && !fused_obs) {   // same contents, as a launch of its own in front (the step below changes the state)
        rc = ic3_env_observe(env, obs, stream);
        if (rc) return rc;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (!inner) {                                                // one-shot (ic3_env_set_step_events): the step's LAST launch
        ev0 = (hipEvent_t)env->ev_start;
        ev1 = (hipEvent_t)env->ev_stop;
        env->ev_start = env->ev_stop = nullptr;
    }
    if (a.npass > 1) {   // comm_passes > 1 as a loop inside the launch (MP instantiations: split gate product, hid 64 / 128)
        if (!a.l_wp3 || (H != 128 && H != 64))
            return fail(-38, "ic3_policy_step: npasses >= 2 needs gate_split and hid_size 64 / 128 (use one call per pass)");
        if (H == 128)
            return pp ? launch_step<128, IC3_ENV_PP, 1, 1>(a, tiles, lds, s, ev0, ev1) : launch_step<128, IC3_ENV_TJ, 1, 1>(a, tiles, lds, s, ev0, ev1);
        return pp ? launch_step<64, IC3_ENV_PP, 1, 1>(a, tiles, lds, s, ev0, ev1) : launch_step<64, IC3_ENV_TJ, 1, 1>(a, tiles, lds, s, ev0, ev1);
    }
    if (a.l_wp3) {   // gate_split: the gate product on the bf16 matrix cores with exact split products
        if (H == 128)
            rc = pp ? launch_step<128, IC3_ENV_PP, 1>(a, tiles, lds, s, ev0, ev1) : launch_step<128, IC3_ENV_TJ, 1>(a, tiles, lds, s, ev0, ev1);
        else if (H == 64)
            rc = pp ? launch_step<64, IC3_ENV_PP, 1>(a, tiles, lds, s, ev0, ev1) : launch_step<64, IC3_ENV_TJ, 1>(a, tiles, lds, s, ev0, ev1);
        else
            rc = pp ? launch_step<256, IC3_ENV_PP, 1>(a, tiles, lds, s, ev0, ev1) : launch_step<256, IC3_ENV_TJ, 1>(a, tiles, lds, s, ev0, ev1);
    } else if (H == 128)
        rc = pp ? launch_step<128, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<128, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
    else if (H == 64)
        rc = pp ? launch_step<64, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<64, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
    else
        rc = pp ? launch_step<256, IC3_ENV_PP>(a, tiles, lds, s, ev0, ev1) : launch_step<256, IC3_ENV_TJ>(a, tiles, lds, s, ev0, ev1);
    return rc;
}


extern "C" int ic3_env_set_record_out(ic3_env* env, float* gates, float* xh)
{
    if (!env) return fail(-22, "ic3_env_set_record_out: null handle");
    if (xh && !gates) return fail(-22, "ic3_env_set_record_out: xh comes with gates");
    env->gates_out = gates;
    env->xh_out = xh;
    return 0;
}

extern "C" int ic3_env_set_hidden_out(ic3_env* env, float* h_out, float* c_out)
{
    if (!env || (h_out == nullptr) != (c_out == nullptr)) return fail(-22, "ic3_env_set_hidden_out: h_out and c_out come together");
    env->h_out = h_out;
    env->c_out = c_out;
    return 0;
}
