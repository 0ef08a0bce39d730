// gates_bwd.hip — update half (trainer.py:128-225 through torch.nn.LSTMCell's backward): the gate pre-activations of one
// recorded step are RE-COMPUTED ([inp | h_prev] . [W_ih | W_hh]^T + b, comm.py:215) and turned into their gradient in the
// same launch — ic3_lstm_gates_backward.
//
// What it replaces in ic3net_amd/bptt.py: a library GEMM writing gates (R x 4H: 168 MB at PP-hard E = 8192), the pointwise
// ic3_lstm_cell_backward reading them back, and a reduction over its bias partials: 284 + 78 + 26 us per step there.
// Here a workgroup owns 64 rows: their [inp | h_prev] rows go to LDS once (the A operand, as in policy_step_kernel), the
// 2H x 4H weights stream from L2 in the k-major float4-over-the-four-gates layout of ic3_policy_pack through an 8-deep
// register ring, v_mfma_f32_32x32x2_f32 accumulates all four gates of a (row, hidden column) in one lane — exact fp32 —
// and the epilogue applies the cell's derivative with c_prev, dL/dh, dL/dc loaded straight into that lane: the
// pre-activations never exist in memory.  Bound: MFMA (2 * R * 2H * 4H flop / 157.3 TFLOP/s = 137 us at R = 81920,
// H = 128); HBM traffic per row: 2H + 3H floats in, 4H + H out (5 KB at H = 128 -> 0.42 GB per call, 52 us at 8 TB/s,
// overlapped by the second workgroup of the CU).
//
// Layouts: accumulator register `reg` of row tile rt <-> tile row 32 rt + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
// hidden column 32 wave + (lane & 31); gate order i, f, g, o (torch.nn.LSTMCell).  Rows past R read as zeros through the
// buffer descriptors' range check (their gradient is exactly 0) and their stores are dropped by it.
#include <hip/hip_runtime.h>

#include "ic3_common.hpp"

namespace ic3 {

typedef float gb_f32x4 __attribute__((ext_vector_type(4)));
typedef float gb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int gb_u32x4 __attribute__((ext_vector_type(4)));

struct GatesBwdArgs {
    float* xh;             // [R][ldx] = [inp | h_prev]
    const float* h_prev;   // or null.  [R][H]: the h half of a row is read from here and ALSO written into xh (the copy
                           // the caller would otherwise make for the weight-gradient product that follows)
    const float* wq;       // ic3_policy_pack's lstm_wp: Wq[k][c] = float4 over the four gates
    const void* wq3;       // gate_split (policy_step.hip; the default): ic3_policy_pack_split's three bf16 planes, or null = fp32 instruction
    const float* bias;     // [4H] b_ih + b_hh
    const float* c_prev;   // [R][H]
    const float* dh;       // [R][H] dL/dh_t
    const float* dc;       // [R][H] dL/dc_t arriving from step t + 1, or null (zeros)
    float* dgates;         // [R][4H]
    float* dc_prev;        // [R][H] (may be `dc`)
    float* dbias;          // [tiles][4H] column sums of dgates per workgroup, or null
    int ldx, R, accumulate;
    // ic3_lstm_gates_backward_dx (round 5, SPLIT only): the input gradient dgates . [W_ih | W_hh] of the same tile in the same
    // launch — wb3 = ic3_policy_pack_split_bwd's planes, dxh [R][2H] = [d inp | d h_prev]; null: dgates only
    const void* wb3;
    float* dxh;
    // ic3_lstm_gates_backward_given (round 5, GIVEN instantiation): the ACTIVATED gates i | f | g | o [R][4H] as the rollout's
    // step launch recorded them (ic3_env_set_gates_out) — no gate product here, only the cell's derivative (+ dx)
    const float* gates;
    // GIVEN, collection mode (trainer.py:227-242 through :128-225): row_live [R] — 0 for the rows of an env that STARTS an episode
    // at this slot (the state that entered it counts as zero: c_prev and the h_prev copied into xh are multiplied by it);
    // row_keep [R] — 0 where the gradient arriving from the next slot must not cross (episode end / detach point): dc times it.
    // null = all ones
    const float* row_live;
    const float* row_keep;
    // GIVEN: the heads' share of dL/dh_t added on the way in (trainer.py:128-225 through comm.py:228,239): dh + dhead . w_heads
    // with dhead [R][OT] = dL/d[logits of every head | value] of the step and w_heads [OT][H] (heads.k.weight stacked, then
    // value_head.weight), OT <= 16 — replaces the R x OT x H library product (and its pass over dh) in front of the launch.
    // null = dh as it is
    const float* dhead;
    const float* w_heads;
    int OT;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gb_rsrc(const void* base, long long bytes)
{
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ gb_f32x4 gb_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(gb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float gb_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void gb_store1(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
// (Cache policy of the GIVEN instantiation's streams, round 5: non-temporal loads / stores measured 94.7 / 94.2 / 94.0 / 93.5 M
// agent-steps/s for none / loads / stores / both inside a PP-hard update — profiles/r05/gates_given_nt_ab.txt — default policy.)

// Matrix instructions through the compiler's builtins ONLY.  Rounds 3-5 pinned the accumulators to AGPRs with inline-asm
// MFMAs (266 instead of 280 us per recompute call); the hazard recogniser cannot see through inline asm, and a split plane
// written by v_cvt_pk_bf16_f32 a few cycles before the asm MFMA that read it arrived stale on warm launches (round 5: errors
// of 3e-6 instead of 9e-7, run-to-run differences, held off by s_nop padding).  With the builtins the compiler inserts the
// wait states itself: safe by construction (round-5 verdict, item 2c; the 5 % are what it costs).
__device__ __forceinline__ void gb_mfma(gb_f32x16& acc, float x, float y)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
}

// SPLIT = 1 (the default; the update half's twin of policy_step_kernel's gate_split): the recompute as nine exact
// bf16 x bf16 products per 16 k-steps on v_mfma_f32_32x32x16_bf16 — the same loop: weight planes in fragment order, one
// 16-k block of them in registers, activations split per wave from the fp32 LDS tile through v_cvt_pk_bf16_f32.
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gb_split_pair(gb_f32x2 x, unsigned& p1, unsigned& p2, unsigned& p3)
{
    const gb_bf16x2 h1 = __builtin_convertvector(x, gb_bf16x2);
    const gb_f32x2 r1 = x - __builtin_convertvector(h1, gb_f32x2);
    const gb_bf16x2 h2 = __builtin_convertvector(r1, gb_bf16x2);
    const gb_f32x2 r2 = r1 - __builtin_convertvector(h2, gb_f32x2);
    const gb_bf16x2 h3 = __builtin_convertvector(r2, gb_bf16x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, h2);
    p3 = __builtin_bit_cast(unsigned, h3);
}
__device__ __forceinline__ void gb_split_frag(gb_f32x4 x0, gb_f32x4 x1, gb_u32x4 (&out)[3])
{
    unsigned p[3][4];
    gb_split_pair(gb_f32x2{ x0[0], x0[1] }, p[0][0], p[1][0], p[2][0]);
    gb_split_pair(gb_f32x2{ x0[2], x0[3] }, p[0][1], p[1][1], p[2][1]);
    gb_split_pair(gb_f32x2{ x1[0], x1[1] }, p[0][2], p[1][2], p[2][2]);
    gb_split_pair(gb_f32x2{ x1[2], x1[3] }, p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = gb_u32x4{ p[pl][0], p[pl][1], p[pl][2], p[pl][3] };
}
__device__ __forceinline__ void gb_mfma_bf16(gb_f32x16& acc, gb_u32x4 x, gb_u32x4 y)
{
    typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gb_bf16x8, x), __builtin_bit_cast(gb_bf16x8, y), acc, 0, 0, 0);
}

template <int H, int SPLIT = 0, int GIVEN = 0>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 2 : 1) void lstm_gates_bwd_kernel(const GatesBwdArgs a)
{
    static_assert(!GIVEN || SPLIT == 1, "recorded gates: the split instantiation (its dx product)");
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, NT = 2 * H, KB = K / 8, K4 = K / 4, RING = 8;
    static_assert(KB % 2 == 0 && KB >= 4, "K / 8 must be even");
    IC3_DYNAMIC_LDS(float, smem);
    float* const As = smem;                                      // [64][LDA]
    gb_f32x4* const As4 = reinterpret_cast<gb_f32x4*>(smem);
    float* const slb = As + 64 * LDA;                            // [4H]
    float* const sdh = slb + 4 * H;                              // GIVEN: [64][16] the tile's dhead rows, zero beyond OT
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int col = 32 * w + li;
    const long long r0 = (long long)blockIdx.x * 64;
    const int rows = (int)((a.R - r0) < 64 ? (a.R - r0) : 64);

    // ---- A tile: 64 rows of [inp | h_prev] -> LDS (all loads issued, then the LDS writes) ------------------------------
    if constexpr (GIVEN != 0) {
        if (a.dhead) {                                           // (uniform) 64 x 16 floats, 4 per thread at H = 128
            const __amdgpu_buffer_rsrc_t rdd = gb_rsrc(a.dhead + r0 * a.OT, (long long)rows * a.OT * 4);
            for (int idx = tid; idx < 64 * 16; idx += NT) {
                const int row = idx >> 4, o = idx & 15;
                sdh[idx] = o < a.OT ? gb_load1(rdd, (row * a.OT + o) * 4, 0) : 0.0f;   // (rows past R: dropped loads read 0)
            }
        }
        // (recorded gates: nothing of the forward runs again — only the caller's copy of h_prev into the h half of xh, the
        //  weight-gradient product's operand, is made here)
        if (a.xh && a.h_prev) {
            const __amdgpu_buffer_rsrc_t rx = gb_rsrc(a.xh + r0 * a.ldx, ((long long)(rows - 1) * a.ldx + K) * 4);
            const __amdgpu_buffer_rsrc_t rhp = gb_rsrc(a.h_prev + r0 * H, (long long)rows * H * 4);
            constexpr int H4 = H / 4, PERH = 64 * H4 / NT;           // float4 per thread (8)
#pragma unroll
            for (int i = 0; i < PERH; ++i) {
                const int idx = tid + i * NT, row = idx / H4, c4 = idx - row * H4;
                gb_f32x4 v = gb_load4(rhp, (row * H + 4 * c4) * 4, 0);
                if (a.row_live) v *= (row < rows ? a.row_live[r0 + row] : 0.0f);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gb_u32x4, v), rx, (row * a.ldx + H + 4 * c4) * 4, 0, 0);
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rx = gb_rsrc(a.xh + r0 * a.ldx, ((long long)(rows - 1) * a.ldx + K) * 4);
        constexpr int PER = 64 * K4 / NT;                        // float4 per thread (16)
        const __amdgpu_buffer_rsrc_t rhp = gb_rsrc(a.h_prev ? a.h_prev + r0 * H : a.xh, a.h_prev ? (long long)rows * H * 4 : 0);
        const bool from_h = a.h_prev != nullptr;
        gb_f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * NT, row = idx / K4, c4 = idx - row * K4;
            if (from_h && c4 >= H / 4) {
                v[i] = gb_load4(rhp, (row * H + 4 * c4 - H) * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gb_u32x4, v[i]), rx, (row * a.ldx + 4 * c4) * 4, 0, 0);
            } else {
                v[i] = gb_load4(rx, (row * a.ldx + 4 * c4) * 4, 0);
            }
        }
        if (tid < H) reinterpret_cast<gb_f32x4*>(slb)[tid] = reinterpret_cast<const gb_f32x4*>(a.bias)[tid];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * NT, row = idx / K4, c4 = idx - row * K4;
            As4[row * LDA4 + c4] = v[i];
        }
    }
    // ---- c_prev of this lane's 32 (row, column) elements: requested now, used behind the gate loop ----------------------
    const int voff = (4 * lh * H + col) * 4;
    float cold[2][16];
    const __amdgpu_buffer_rsrc_t rc = gb_rsrc(a.c_prev + r0 * H, (long long)rows * H * 4);
    if constexpr (SPLIT == 0) {   // (split loop: 48 plane + 24 activation registers next to them — requested behind the loop there)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                cold[rt][reg] = gb_load1(rc, voff, (32 * rt + (reg & 3) + 8 * (reg >> 2)) * H * 4);
    }
    gb_f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][gt][i] = 0.0f;
    if constexpr (GIVEN != 0) {
    } else if constexpr (SPLIT != 0) {
        constexpr int KB16 = K / 16, NWv = H / 32;
        const __amdgpu_buffer_rsrc_t rg3 = gb_rsrc(a.wq3, (long long)3 * K * 4 * H * 2);
        const int g3lane = (w * 64 + lane) * 16;
        constexpr int GSTRIDE = NWv * 64 * 16;
        auto wq3 = [&](int pl, int kb, int gt) {
            return __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, ((pl * KB16 + kb) * 4 + gt) * GSTRIDE, 0);
        };
        gb_u32x4 bq[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) bq[pl][gt] = wq3(pl, 0, gt);
        __syncthreads();
        auto block3 = [&](auto refill_c, int kb) {
            constexpr bool REFILL = decltype(refill_c)::value;
            gb_u32x4 ap[2][3];
            {
                const gb_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh;
                const gb_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh;
                gb_split_frag(s0[0], s0[1], ap[0]);
                gb_split_frag(s1[0], s1[1], ap[1]);
            }
#pragma unroll
            for (int pb = 0; pb < 3; ++pb)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {                     // the six products of one weight fragment, then its refill
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa) {
                        gb_mfma_bf16(acc[0][gt], ap[0][pa], bq[pb][gt]);
                        gb_mfma_bf16(acc[1][gt], ap[1][pa], bq[pb][gt]);
                    }
                    if constexpr (REFILL) bq[pb][gt] = wq3(pb, kb + 1, gt);
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
#pragma unroll 1
        for (int kb = 0; kb < KB16 - 1; ++kb) block3(std::true_type{}, kb);
        block3(std::false_type{}, KB16 - 1);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        // ---- gates = [inp | h] . [W_ih | W_hh]^T: k = 8 kb + 4 lh + j (A fragment and B slot agree) --------------------------
        const __amdgpu_buffer_rsrc_t rgw = gb_rsrc(a.wq, (long long)K * 4 * H * 4);
        const int glane = (4 * lh * H + col) * 16;
        auto wq = [&](int kb, int j) { return gb_load4(rgw, glane, (8 * kb + j) * (H * 16)); };
        gb_f32x4 wk[RING];
    #pragma unroll
        for (int i = 0; i < RING; ++i) wk[i] = wq(i >> 2, i & 3);
        __syncthreads();
        auto block = [&](auto sb_c, auto refill_c, int kb) {
            constexpr int SB = decltype(sb_c)::value;
            constexpr bool REFILL = decltype(refill_c)::value;
            const gb_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
            const gb_f32x4 a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
    #pragma unroll
                for (int gt = 0; gt < 4; ++gt) {
                    gb_mfma(acc[0][gt], a0[j], wk[SB + j][gt]);
                    gb_mfma(acc[1][gt], a1[j], wk[SB + j][gt]);
                }
                if constexpr (REFILL) wk[SB + j] = wq(kb + RING / 4, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr std::integral_constant<int, 0> s0{};
        constexpr std::integral_constant<int, 4> s1{};
    #pragma unroll 1
        for (int kb = 0; kb < KB - 2; kb += 2) {
            block(s0, std::true_type{}, kb);
            block(s1, std::true_type{}, kb + 1);
        }
        block(s0, std::false_type{}, KB - 2);
        block(s1, std::false_type{}, KB - 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- the cell's derivative (torch.nn.LSTMCell): c' = f c + i g, h' = o tanh(c') -----------------------------------------
    const float bi = GIVEN ? 0.f : slb[col], bf = GIVEN ? 0.f : slb[H + col], bg = GIVEN ? 0.f : slb[2 * H + col], bo = GIVEN ? 0.f : slb[3 * H + col];
    const bool dx = SPLIT != 0 && a.dxh != nullptr;                  // (uniform) the input gradient in this launch too
    float keep_g[2][16], keep_o[2][16];
    const bool heads_in = GIVEN != 0 && a.dhead != nullptr;          // (uniform)
    if (dx || heads_in) __syncthreads();                             // every wave is done reading the A tile: it takes d i, d f now
    float wh[16] = {};                                               // w_heads[o][col] of this lane's hidden column
    if constexpr (GIVEN != 0) {
        if (heads_in) {
            const __amdgpu_buffer_rsrc_t rwh = gb_rsrc(a.w_heads, (long long)a.OT * H * 4);
#pragma unroll
            for (int o = 0; o < 16; ++o) wh[o] = gb_load1(rwh, (o * H + col) * 4, 0);     // (o >= OT: out of range, 0)
        }
    }
    const long long nrec = (long long)rows * H * 4;
    // (dh / dc null: every load reads 0 through an empty descriptor — a detach point of the recurrence costs no memset and no read)
    const __amdgpu_buffer_rsrc_t rdh = gb_rsrc(a.dh ? a.dh + r0 * H : a.c_prev, a.dh ? nrec : 0);
    const __amdgpu_buffer_rsrc_t rdc = gb_rsrc(a.dc ? a.dc + r0 * H : a.c_prev, a.dc ? nrec : 0);
    const __amdgpu_buffer_rsrc_t rdp = gb_rsrc(a.dc_prev + r0 * H, nrec);
    const __amdgpu_buffer_rsrc_t rdg = gb_rsrc(a.dgates + r0 * 4 * H, 4 * nrec);
    const int goff = (4 * lh * 4 * H + col) * 4;
    const __amdgpu_buffer_rsrc_t rgin = gb_rsrc(GIVEN ? a.gates + r0 * 4 * H : a.c_prev, GIVEN ? 4 * nrec : 0);
    const bool cuts = GIVEN != 0 && (a.row_live != nullptr || a.row_keep != nullptr);
    const __amdgpu_buffer_rsrc_t rlive = gb_rsrc(a.row_live ? a.row_live + r0 : a.c_prev, a.row_live ? (long long)rows * 4 : 0);
    const __amdgpu_buffer_rsrc_t rkeep = gb_rsrc(a.row_keep ? a.row_keep + r0 : a.c_prev, a.row_keep ? (long long)rows * 4 : 0);
    float si = 0.f, sf = 0.f, sg = 0.f, so = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        float dhv[16], dcv[16], gin[GIVEN ? 4 : 1][16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
            dhv[reg] = gb_load1(rdh, voff, lc * H * 4);
            dcv[reg] = gb_load1(rdc, voff, lc * H * 4);
            if constexpr (SPLIT != 0) cold[rt][reg] = gb_load1(rc, voff, lc * H * 4);
            if constexpr (GIVEN != 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) gin[k][reg] = gb_load1(rgin, goff, lc * 4 * H * 4 + k * H * 4);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
            float i, f, gt, o;
            if constexpr (GIVEN != 0) {
                i = gin[0][reg], f = gin[1][reg], gt = gin[2][reg], o = gin[3][reg];
                if (heads_in) {                                      // dL/dh_t += dhead[row] . w_heads[:, col]
                    const gb_f32x4* dr = reinterpret_cast<const gb_f32x4*>(sdh + (lc + 4 * lh) * 16);   // (a broadcast read)
                    float s = dhv[reg];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (4 * q >= a.OT) break;
                        const gb_f32x4 dv = dr[q];
                        s = __builtin_fmaf(dv[0], wh[4 * q], s);
                        s = __builtin_fmaf(dv[1], wh[4 * q + 1], s);
                        s = __builtin_fmaf(dv[2], wh[4 * q + 2], s);
                        s = __builtin_fmaf(dv[3], wh[4 * q + 3], s);
                    }
                    dhv[reg] = s;
                }
                if (cuts) {                                          // (uniform) collection mode: the slot's per-row cuts
                    const int lrow = lc + 4 * lh;
                    if (a.row_live) cold[rt][reg] *= gb_load1(rlive, lrow * 4, 0);
                    if (a.row_keep) dcv[reg] *= gb_load1(rkeep, lrow * 4, 0);
                }
            } else {
                i = fast_sigmoid(acc[rt][0][reg] + bi), f = fast_sigmoid(acc[rt][1][reg] + bf);
                gt = fast_tanh(acc[rt][2][reg] + bg), o = fast_sigmoid(acc[rt][3][reg] + bo);
            }
            const float c0 = cold[rt][reg];
            const float tc = fast_tanh(__builtin_fmaf(f, c0, i * gt));   // (the forward's own rounding: policy_step.hip's cell)
            const float dct = dcv[reg] + dhv[reg] * o * (1.0f - tc * tc);
            const float di = dct * gt * i * (1.0f - i), df = dct * c0 * f * (1.0f - f);
            const float dg = dct * i * (1.0f - gt * gt), dO = dhv[reg] * tc * o * (1.0f - o);
            gb_store1(di, rdg, goff, lc * 4 * H * 4);
            gb_store1(df, rdg, goff, lc * 4 * H * 4 + H * 4);
            gb_store1(dg, rdg, goff, lc * 4 * H * 4 + 2 * H * 4);
            gb_store1(dO, rdg, goff, lc * 4 * H * 4 + 3 * H * 4);
            gb_store1(dct * f, rdp, voff, lc * H * 4);
            if constexpr (SPLIT != 0) {
                if (dx) {                                            // K-half 0 of dgates (gates i, f) -> the A tile; g, o wait in registers
                    const int lr = lc + 4 * lh;
                    As[lr * LDA + col] = di;
                    As[lr * LDA + H + col] = df;
                    keep_g[rt][reg] = dg;
                    keep_o[rt][reg] = dO;
                }
            }
            si += di;
            sf += df;
            sg += dg;
            so += dO;
        }
    }
    if constexpr (SPLIT != 0) {
        if (dx) {
            // ---- [d inp | d h_prev] = dgates . [W_ih | W_hh]  (64 x 4H . 4H x 2H): the tile's dgates are the A operand from LDS
            // (fp32, split per wave like the gate loop's), in two K halves through the A tile's 2H columns; the weights: three
            // bf16 planes in fragment order (ic3_policy_pack_split_bwd), wave w owns output columns [64 w, 64 w + 64).  Nine
            // exact products per fp32 product, fp32 accumulation — the arithmetic of the gate product.
            constexpr int KB16B = 4 * H / 16, NCT = 2 * H / 32, KBH = K / 16;
            const __amdgpu_buffer_rsrc_t rwb = gb_rsrc(a.wb3, (long long)3 * 4 * H * 2 * H * 2);
            const int wu = __builtin_amdgcn_readfirstlane(w);    // (scalar offset: no waterfall loop around each load)
            auto wb = [&](int pl, int kg, int ct) __attribute__((always_inline)) {
                return __builtin_amdgcn_raw_buffer_load_b128(rwb, lane * 16, ((pl * KB16B + kg) * NCT + 2 * wu + ct) * 1024, 0);
            };
            gb_u32x4 bq2[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bq2[pl][ct] = wb(pl, 0, ct);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.0f;
            auto blockdx = [&](int kb, int kg) __attribute__((always_inline)) {
                gb_u32x4 ap[2][3];
                const gb_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh;
                const gb_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh;
                gb_split_frag(s0[0], s0[1], ap[0]);
                gb_split_frag(s1[0], s1[1], ap[1]);
#pragma unroll
                for (int pb = 0; pb < 3; ++pb)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                        for (int pa = 2; pa >= 0; --pa) {
                            gb_mfma_bf16(acc[0][ct], ap[0][pa], bq2[pb][ct]);
                            gb_mfma_bf16(acc[1][ct], ap[1][pa], bq2[pb][ct]);
                        }
                        if (kg + 1 < KB16B) bq2[pb][ct] = wb(pb, kg + 1, ct);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            };
            __syncthreads();                                     // d i, d f of every wave are in the A tile
#pragma unroll 1
            for (int kb = 0; kb < KBH; ++kb) blockdx(kb, kb);
            __syncthreads();                                     // every wave is done with K-half 0
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    As[lr * LDA + col] = keep_g[rt][reg];
                    As[lr * LDA + H + col] = keep_o[rt][reg];
                }
            __syncthreads();
#pragma unroll 1
            for (int kb = 0; kb < KBH; ++kb) blockdx(kb, KBH + kb);
            const __amdgpu_buffer_rsrc_t rdx = gb_rsrc(a.dxh + r0 * K, (long long)rows * K * 4);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        gb_store1(acc[rt][ct][reg], rdx, (lr * K + 64 * w + 32 * ct + li) * 4, 0);
                    }
        }
    }
    if (a.dbias) {                                               // column sums over the tile's 64 rows: the two lane halves
        si += __shfl_xor(si, 32);
        sf += __shfl_xor(sf, 32);
        sg += __shfl_xor(sg, 32);
        so += __shfl_xor(so, 32);
        if (lh == 0) {
            float* d = a.dbias + (size_t)blockIdx.x * 4 * H + col;
            if (a.accumulate) {
                d[0] += si;
                d[H] += sf;
                d[2 * H] += sg;
                d[3 * H] += so;
            } else {
                d[0] = si;
                d[H] = sf;
                d[2 * H] = sg;
                d[3 * H] = so;
            }
        }
    }
}

}  // namespace ic3

extern "C" int ic3_lstm_gates_backward_supported(int H) { return H == 64 || H == 128 || H == 256; }

static int gates_backward_impl(float* xh, int ldx, const float* h_prev, const float* lstm_wp, const void* lstm_wp3, const float* bias,
                               const float* c_prev, const float* dh, const float* dc, float* dgates, float* dc_prev,
                               float* dbias_partials, int accumulate, const void* wb3, float* dxh, int R, int H, ic3_stream stream,
                               const float* gates = nullptr, const float* row_live = nullptr, const float* row_keep = nullptr,
                               const float* dhead = nullptr, const float* w_heads = nullptr, int OT = 0);

extern "C" int ic3_lstm_gates_backward(float* xh, int ldx, const float* h_prev, const float* lstm_wp, const void* lstm_wp3, const float* bias, const float* c_prev,
                                       const float* dh, const float* dc, float* dgates, float* dc_prev, float* dbias_partials,
                                       int accumulate, int R, int H, ic3_stream stream)
{
    return gates_backward_impl(xh, ldx, h_prev, lstm_wp, lstm_wp3, bias, c_prev, dh, dc, dgates, dc_prev, dbias_partials, accumulate,
                               nullptr, nullptr, R, H, stream);
}

extern "C" int ic3_lstm_gates_backward_dx(float* xh, int ldx, const float* h_prev, const float* lstm_wp, const void* lstm_wp3,
                                          const void* lstm_wp3_bwd, const float* bias, const float* c_prev, const float* dh,
                                          const float* dc, float* dgates, float* dc_prev, float* dbias_partials, int accumulate,
                                          float* dxh, int R, int H, ic3_stream stream)
{
    if (!lstm_wp3 || !lstm_wp3_bwd || !dxh)
        return ic3::fail(-22, "ic3_lstm_gates_backward_dx: needs the split planes of both products (lstm_wp3, lstm_wp3_bwd) and dxh");
    if (H != 64 && H != 128) return ic3::fail(-38, "ic3_lstm_gates_backward_dx: hid_size 64 / 128");
    return gates_backward_impl(xh, ldx, h_prev, lstm_wp, lstm_wp3, bias, c_prev, dh, dc, dgates, dc_prev, dbias_partials, accumulate,
                               lstm_wp3_bwd, dxh, R, H, stream);
}

extern "C" int ic3_lstm_gates_backward_given(const float* gates, float* xh, int ldx, const float* h_prev, const void* lstm_wp3_bwd,
                                             const float* c_prev, const float* dh, const float* dc, float* dgates, float* dc_prev,
                                             float* dbias_partials, int accumulate, float* dxh, const float* row_live,
                                             const float* row_keep, const float* dhead, const float* w_heads, int OT, int R, int H,
                                             ic3_stream stream)
{
    if (!gates) return ic3::fail(-22, "ic3_lstm_gates_backward_given: null gates");
    if ((dhead == nullptr) != (w_heads == nullptr) || (dhead && (OT < 1 || OT > 16)))
        return ic3::fail(-22, "ic3_lstm_gates_backward_given: dhead [R][OT] and w_heads [OT][H] come together, 1 <= OT <= 16");
    if (row_keep && !dc) return ic3::fail(-22, "ic3_lstm_gates_backward_given: row_keep scales dc");
    if ((lstm_wp3_bwd == nullptr) != (dxh == nullptr))
        return ic3::fail(-22, "ic3_lstm_gates_backward_given: lstm_wp3_bwd and dxh come together");
    if (H != 64 && H != 128) return ic3::fail(-38, "ic3_lstm_gates_backward_given: hid_size 64 / 128");
    if ((xh == nullptr) != (h_prev == nullptr))
        return ic3::fail(-22, "ic3_lstm_gates_backward_given: xh and h_prev come together (the copy into xh's h half) or not at all");
    return gates_backward_impl(xh, xh ? ldx : 2 * H, h_prev, nullptr, nullptr, nullptr, c_prev, dh, dc, dgates, dc_prev, dbias_partials,
                               accumulate, lstm_wp3_bwd, dxh, R, H, stream, gates, row_live, row_keep, dhead, w_heads, OT);
}

static int gates_backward_impl(float* xh, int ldx, const float* h_prev, const float* lstm_wp, const void* lstm_wp3, const float* bias,
                               const float* c_prev, const float* dh, const float* dc, float* dgates, float* dc_prev,
                               float* dbias_partials, int accumulate, const void* wb3, float* dxh, int R, int H, ic3_stream stream,
                               const float* gates, const float* row_live, const float* row_keep, const float* dhead,
                               const float* w_heads, int OT)
{
    using namespace ic3;
    if ((!gates && (!xh || !lstm_wp || !bias)) || !c_prev || (!dh && !gates) || !dgates || !dc_prev || R <= 0)
        return fail(-22, "ic3_lstm_gates_backward: null argument");
    if (!ic3_lstm_gates_backward_supported(H)) return fail(-38, "ic3_lstm_gates_backward: needs hid_size 64 / 128 / 256");
    if (ldx < 2 * H || (ldx & 3)) return fail(-22, "ic3_lstm_gates_backward: ldx must be a multiple of 4, >= 2 * hid_size");
    if ((long long)R * (ldx > 4 * H ? ldx : 4 * H) * 4 >= (1ll << 32))
        return fail(-22, "ic3_lstm_gates_backward: R * 4H floats must stay below 4 GB (32-bit buffer offsets)");
    const GatesBwdArgs a{ xh, h_prev, lstm_wp, lstm_wp3, bias, c_prev, dh, dc, dgates, dc_prev, dbias_partials, ldx, R, accumulate, wb3, dxh, gates, row_live, row_keep, dhead, w_heads, OT };
    const int tiles = (R + 63) / 64;
    const size_t lds = ((size_t)64 * (2 * H + 4) + 4 * H + (gates ? 64 * 16 : 0)) * sizeof(float);   // (+ the dhead tile)
    hipStream_t s = (hipStream_t)stream;
    if (gates) {                                                 // (H 64 / 128: checked by the entry point)
        if (H == 128) {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_gates_bwd_kernel<128, 1, 1>), lds));
            hipLaunchKernelGGL((lstm_gates_bwd_kernel<128, 1, 1>), dim3(tiles), dim3(256), lds, s, a);
        } else {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_gates_bwd_kernel<64, 1, 1>), lds));
            hipLaunchKernelGGL((lstm_gates_bwd_kernel<64, 1, 1>), dim3(tiles), dim3(128), lds, s, a);
        }
        IC3_HIP(hipGetLastError());
        return tiles;
    }
#define IC3_GB(h)                                                                                                       \
    case h:                                                                                                             \
        if (lstm_wp3) {                                                                                                 \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_gates_bwd_kernel<h, 1>), lds));               \
            hipLaunchKernelGGL((lstm_gates_bwd_kernel<h, 1>), dim3(tiles), dim3(2 * h), lds, s, a);                     \
        } else {                                                                                                        \
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_gates_bwd_kernel<h, 0>), lds));               \
            hipLaunchKernelGGL((lstm_gates_bwd_kernel<h, 0>), dim3(tiles), dim3(2 * h), lds, s, a);                     \
        }                                                                                                               \
        break;
    switch (H) {
        IC3_GB(64)
        IC3_GB(128)
        IC3_GB(256)
    }
#undef IC3_GB
    IC3_HIP(hipGetLastError());
    return tiles;   // rows of dbias_partials written
}
