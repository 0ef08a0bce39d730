// bptt_kernels.hip — update half (trainer.py:128-225 differentiating comm.py:134-244 over a recorded rollout): what the backward
// through time needs besides the LSTM cell's derivative (gates_bwd.hip), as hand-written launches — no library GEMM is left in
// the per-step chain of the recorded-gates path:
//
//   ic3_comm_backward     one launch per recorded step: the communication block + C's share of dL/dh_{t-1} and C.weight's
//                         gradient.  With M the per-env mixing matrix of comm.py:181-205 (symmetric: ic3_comm_masked_mean),
//                         comm = M h_prev, inp = enc + comm C^T:
//                             d h_prev = d h_direct + M (d inp . C) = d h_direct + (M d inp) . C
//                             d C     += d inp^T . comm             = (M d inp)^T . h_prev
//                         so ONE mix (of d inp, in LDS) feeds both products — the forward's comm is never formed again.  Replaces
//                         two masked-mean launches and two library products per step.
//   ic3_lstm_weight_grad  ONE launch per window of recorded steps: d[W_ih | W_hh]^T += [inp | h_prev]^T . dgates over all
//                         T x R rows at once (split-K over the CUs, fixed-order reduction) — replaces a library product per step.
//   ic3_bptt_backward     the loop over a window's steps, last to first, as ONE host call: cell derivative + input gradient
//                         (ic3_lstm_gates_backward_given, the heads' share of dL/dh folded in) -> ic3_comm_backward -> the sparse
//                         encoder's backward stage 1 — three launches per step, no host work between them.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation — the library products these replace ran on
// the same instruction).  Layouts as gates_bwd.hip: accumulator register `reg` of a 32 x 32 block <-> block row
// (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), block column lane & 31.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "enc_bwd.hpp"
#include "ic3_common.hpp"
#include "ps_common.hpp"

extern "C" int ic3_lstm_gates_backward_given(const float* gates, float* xh, int ldx, const float* h_prev, const void* lstm_wp3_bwd,
                                             const float* c_prev, const float* dh, const float* dc, float* dgates, float* dc_prev,
                                             float* dbias_partials, int accumulate, float* dxh, const float* row_live,
                                             const float* row_keep, const float* dhead, const float* w_heads, int OT, int R, int H,
                                             ic3_stream stream);

namespace ic3 {

typedef float bp_f32x2 __attribute__((ext_vector_type(2)));
typedef float bp_f32x4 __attribute__((ext_vector_type(4)));
typedef float bp_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bp_rsrc(const void* base, long long bytes)
{
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float bp_load1(__amdgpu_buffer_rsrc_t r, int voff, int soff = 0)   // (soff: the wave-uniform part)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ bp_f32x4 bp_load4(__amdgpu_buffer_rsrc_t r, int voff)
{
    return __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void bp_mfma(bp_f32x16& acc, float x, float y)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// ic3_comm_backward.  A workgroup walks tiles of `ept` whole envs (<= 64 agent rows, the tiling of policy_step_kernel); wave w
// owns hidden columns [32 w, 32 w + 32).  Per tile:
//   0. d inp rows -> LDS, mixed in place: m_j = g_j (S - g_j x_j) scale with g = alive * gate, S = sum_i g_i x_i (the closed
//      form of ic3_comm_masked_mean, same expression order)
//   1. d h_direct + m . C (64 x H x H; d h_direct is loaded INTO the accumulators), epilogue: dh_out = that * out_scale
//   2. dC[k][n] += sum_rows m[row][k] h_prev[row][n] — accumulators live across the workgroup's tiles, one partial per workgroup
// HBM per agent row: d inp, d h_direct, h_prev in, dh_out out = 4 H floats (2 KB at H = 128); MFMA: 2 x 2 H^2 flop per row.
// ---------------------------------------------------------------------------------------------------------------------------
struct CommBwdArgs {
    const float* dxh;        // [R][ldd]: columns [0, H) = d inp, [H, 2H) = d h_prev of the gate product
    const float* h_prev;     // [R][H]
    const int32_t* alive;    // [E][N] or null (everyone: quirk Q21)
    const int32_t* gate;     // [E][N] or null (everyone talks)
    const float* cw;         // C.weight [H][H] (out, in): d comm = d inp . cw
    const float* out_scale;  // [R] or null: dh_out rows times it (collection mode: the gradient that must not cross a cut)
    float* dh_out;           // [R][H]
    float* dcw_part;         // [gridDim.x][H][H]
    int ldd, E, N, ept, tiles, mode_avg, accumulate;
};

template <int H>
__global__ __launch_bounds__(2 * H, (H <= 128) ? 3 : 1) void comm_bwd_kernel(const CommBwdArgs a)
{
    constexpr int NT = 2 * H, H4 = H / 4, LDA = H + 4, LDA4 = LDA / 4, MB = H / 32, PER = 64 * H4 / NT;
    IC3_DYNAMIC_LDS(float, smem);
    float* const Am = smem;                                      // [64][LDA]
    bp_f32x4* const Am4 = reinterpret_cast<bp_f32x4*>(smem);
    float* const sg = smem + 64 * LDA;                           // [64] alive * gate of the tile's rows
    float* const sal = sg + 64;                                  // [64] alive
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int col = 32 * w + li;
    const int N = a.N;
    bp_f32x16 acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[mb][i] = 0.0f;
    const __amdgpu_buffer_rsrc_t rcw = bp_rsrc(a.cw, (long long)H * H * 4);

    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int e0 = tile * a.ept;
        const int ne = (a.E - e0) < a.ept ? (a.E - e0) : a.ept;
        const int rows = ne * N;
        const long long r0 = (long long)e0 * N;
        // ---- every load of the tile is requested here: d inp rows (-> LDS), d h_direct straight into phase 1's accumulators
        // (P is added on top of it), the masks, h_prev of this lane's (row pair, column) for phase 2.  Rows past the tile read 0.
        const __amdgpu_buffer_rsrc_t rdi = bp_rsrc(a.dxh + r0 * a.ldd, ((long long)(rows - 1) * a.ldd + H) * 4);
        const __amdgpu_buffer_rsrc_t rdd = bp_rsrc(a.dxh + r0 * a.ldd + H, ((long long)(rows - 1) * a.ldd + H) * 4);
        const __amdgpu_buffer_rsrc_t rhp = bp_rsrc(a.h_prev + r0 * H, (long long)rows * H * 4);
        bp_f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * NT, row = idx / H4, c4 = idx - row * H4;
            v[i] = bp_load4(rdi, (row * a.ldd + 4 * c4) * 4);
        }
        float mg = 0.f, mal = 0.f;
        if (tid < 64 && tid < rows) {
            mal = a.alive ? (float)a.alive[r0 + tid] : 1.0f;                                     // quirk Q21: no mask = everyone
            mg = mal * (a.gate ? (float)a.gate[r0 + tid] : 1.0f);
        }
        bp_f32x16 acc1[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                acc1[rt][reg] = bp_load1(rdd, (4 * lh * a.ldd + col) * 4, (32 * rt + (reg & 3) + 8 * (reg >> 2)) * a.ldd * 4);
#pragma unroll
        for (int i = 0; i < PER; ++i) Am4[(tid + i * NT) / H4 * LDA4 + (tid + i * NT) % H4] = v[i];
        if (tid < 64) {
            sg[tid] = mg;
            sal[tid] = mal;
        }
        float hpv[32];                                           // (behind the staging registers: in flight during phases 0 and 1)
#pragma unroll
        for (int s = 0; s < 32; ++s) hpv[s] = bp_load1(rhp, (lh * H + col) * 4, 2 * s * H * 4);
        __syncthreads();
        // ---- phase 0: the rows mixed per env, in place in LDS ------------------------------------------------------------
        for (int item = tid; item < ne * H4; item += NT) {
            const int el = item / H4, c4 = item - el * H4;
            float na = 0.f;
            for (int j = 0; j < N; ++j) na += sal[el * N + j];                                   // comm.py:102-107
            const int n_alive = (int)na;
            const float scale = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;   // comm.py:194-196, Q23
            bp_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
            for (int i = 0; i < N; ++i) S = mask_fma4(sg[el * N + i], Am4[(el * N + i) * LDA4 + c4], S);   // (ic3_common.hpp: not packed)
            for (int j = 0; j < N; ++j) {
                const float m = sg[el * N + j];
                const bp_f32x4 x = Am4[(el * N + j) * LDA4 + c4];
                Am4[(el * N + j) * LDA4 + c4] = comm_out4(m, S, x, scale);
            }
        }
        __syncthreads();
        // ---- phase 1: d h_direct + m . C  (k = 8 kb + 4 lh + j: A fragment and B slot agree) ---------------------------------
        float wv[4], wn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = bp_load1(rcw, (4 * lh * H + col) * 4, j * H * 4);
#pragma unroll 2
        for (int kb = 0; kb < H / 8; ++kb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wn[j] = bp_load1(rcw, (4 * lh * H + col) * 4, (8 * (kb + 1) + j) * H * 4);   // (past the end: 0)
            const bp_f32x4 a0 = Am4[li * LDA4 + 2 * kb + lh];
            const bp_f32x4 a1 = Am4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bp_mfma(acc1[0], a0[j], wv[j]);
                bp_mfma(acc1[1], a1[j], wv[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = wn[j];
        }
        {
            const __amdgpu_buffer_rsrc_t rout = bp_rsrc(a.dh_out + r0 * H, (long long)rows * H * 4);
            const __amdgpu_buffer_rsrc_t rsc = bp_rsrc(a.out_scale ? a.out_scale + r0 : a.dh_out, a.out_scale ? (long long)rows * 4 : 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                    float o = acc1[rt][reg];
                    if (a.out_scale) o *= bp_load1(rsc, 4 * lh * 4, lc * 4);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), rout, (4 * lh * H + col) * 4, lc * H * 4, 0);   // (past the tile: dropped)
                }
        }
        // ---- phase 2: dC[k][n] += sum_rows m[row][k] h_prev[row][n]: A = m^T from LDS, B = the h_prev registers ---------------
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) bp_mfma(acc2[mb], Am[(2 * s + lh) * LDA + 32 * mb + li], hpv[s]);
        }
        __syncthreads();                                         // every wave is done with the tile
    }
    float* dst = a.dcw_part + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int k = 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (a.accumulate) dst[k * H + col] += acc2[mb][reg];
            else dst[k * H + col] = acc2[mb][reg];
        }
}

// comm_mask_zero (comm.py:40-41: C sees zeros — also the IRIC stand-in): dL/dh_{t-1} is the gate product's share alone
__global__ __launch_bounds__(256) void dh_copy_kernel(const float* __restrict__ dxh, int ldd, const float* __restrict__ out_scale,
                                                      float* __restrict__ dh_out, long long R, int H4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < R * H4; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / H4;
        const int c4 = (int)(i - row * H4);
        bp_f32x4 v = *reinterpret_cast<const bp_f32x4*>(dxh + row * ldd + 4 * H4 + 4 * c4);
        if (out_scale) v *= out_scale[row];
        reinterpret_cast<bp_f32x4*>(dh_out)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ic3_lstm_weight_grad.  dW[m][n] = sum_q X[q][m] D[q][n], X = [inp | h_prev] (2H columns), D = dgates (4H columns), q over
// the Q = T x R recorded rows.  Workgroup (ks, ny): K-slice ks of the rows, output columns [128 ny, 128 ny + 128), all 2H output
// rows; 4 waves as 2 (m) x 2 (n), a wave holds H x 64 of the result (128 accumulator registers at H = 128); two workgroups per
// CU, so that one's staging and barrier sit beside the other's matrix work (a single 8-wave workgroup per CU ran in lock-step:
// 106 instead of ... TFLOP/s).
// Both operands are row-major with q outermost — exactly what v_mfma_f32_32x32x2_f32 wants of a K-major pair: lane (i, kk)
// supplies X[q0 + kk][m(i)] and D[q0 + kk][n(i)], consecutive lanes read consecutive floats, no transposes anywhere.  One
// ds_read_b128 of X feeds the A operands of four m-blocks (block j takes element j: m = 4 i + j — a permutation of the output
// rows the epilogue undoes), one ds_read_b64 of D the B operands of two n-blocks: 6 LDS dwords per 8 MFMAs.
// Staging: KT = 16 rows per stage, global -> registers -> LDS, double-buffered, one barrier per stage.
// Bound: MFMA (2 Q 2H 4H flop on the fp32 instruction: 1.72 TFLOP for a PP-hard update); X is read once per column block
// (4 H / 128 times, from L2 when the blocks of a slice run together: they are gridDim.x apart, i.e. on one XCD), D once.
// ---------------------------------------------------------------------------------------------------------------------------
struct WGradArgs {
    const float* inp;        // [Q][ldi]: the first H floats of a row = inp
    const float* h;          // [Q][H]   h_prev
    const float* dg;         // [Q][4H]  dgates
    const float* row_live;   // [Q] or null: h rows times it
    float* part;             // [gridDim.x][2H][4H]
    long long Q;
    int ldi, rows_per_wg;    // rows per K slice (a multiple of 16)
};

template <int H>
__global__ __launch_bounds__(256, 2) void lstm_wgrad_kernel(const WGradArgs a)
{
    constexpr int KT = 16, XW = 2 * H, DW = 128, MB = H / 32, NB = 2, NT = 256;
    constexpr int X4R = XW / 4, D4R = DW / 4;                    // float4 per staged row
    constexpr int XPT = KT * X4R / NT, DPT = KT * D4R / NT;      // float4 per thread and stage (4 / 2 at H = 128)
    static_assert(XPT >= 1 && DPT == 2 && (NT % X4R) == 0, "staging split");
    IC3_DYNAMIC_LDS(float, smem);
    constexpr int SW = KT * (XW + DW);                           // stage b: X at smem + b * SW, D behind it
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = w & 1, wn = w >> 1;
    const int ny = blockIdx.y;
    const long long q0 = (long long)blockIdx.x * a.rows_per_wg;
    long long nq = a.Q - q0;
    if (nq > a.rows_per_wg) nq = a.rows_per_wg;
    if (nq < 0) nq = 0;
    const __amdgpu_buffer_rsrc_t ri = bp_rsrc(a.inp + q0 * a.ldi, nq > 0 ? ((nq - 1) * a.ldi + H) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rh = bp_rsrc(a.h + q0 * H, nq * H * 4);
    const __amdgpu_buffer_rsrc_t rd = bp_rsrc(a.dg + q0 * 4 * H + ny * DW, nq > 0 ? ((nq - 1) * 4 * H + DW) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rl = bp_rsrc(a.row_live ? a.row_live + q0 : a.h, a.row_live ? nq * 4 : 0);
    const int nstages = (int)((nq + KT - 1) / KT);
    // a thread stages the same (row-in-stage, column chunk) of every stage: X chunk i at row xrow + i * (NT / X4R)
    const int xrow = tid / X4R, xc4 = tid - xrow * X4R;
    const bool x_is_h = xc4 >= H / 4;
    const int xvoff = x_is_h ? (xrow * H + 4 * xc4 - H) * 4 : (xrow * a.ldi + 4 * xc4) * 4;
    const int xstep = (NT / X4R) * (x_is_h ? H : a.ldi) * 4;    // bytes between two of the thread's chunks
    const int drow = tid / D4R, dc4 = tid - drow * D4R;
    const int dvoff = (drow * 4 * H + 4 * dc4) * 4;

    bp_f32x4 xr[XPT], dr[DPT];
    auto fetch = [&](int s) {
        const int qb = s * KT;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            if (x_is_h) {
                xr[i] = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, xvoff + i * xstep, qb * H * 4, 0));
                if (a.row_live) xr[i] *= bp_load1(rl, (xrow + i * (NT / X4R)) * 4, qb * 4);
            } else {
                xr[i] = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, xvoff + i * xstep, qb * a.ldi * 4, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < DPT; ++i)
            dr[i] = __builtin_bit_cast(bp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, dvoff + i * (NT / D4R) * 4 * H * 4, qb * 4 * H * 4, 0));
    };
    auto stash = [&](int b) {
        bp_f32x4* X4 = reinterpret_cast<bp_f32x4*>(smem + b * SW);
        bp_f32x4* D4 = reinterpret_cast<bp_f32x4*>(smem + b * SW + KT * XW);
#pragma unroll
        for (int i = 0; i < XPT; ++i) X4[tid + i * NT] = xr[i];
#pragma unroll
        for (int i = 0; i < DPT; ++i) D4[tid + i * NT] = dr[i];
    };
    bp_f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;
    if (nstages > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const bool more = s + 1 < nstages;
        if (more) fetch(s + 1);
        const float* Xs = smem + (s & 1) * SW;
        const float* Ds = Xs + KT * XW;
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            const int kr = 2 * ks + lh;
            float av[MB];
            if constexpr (MB == 4) {
                const bp_f32x4 t4 = *reinterpret_cast<const bp_f32x4*>(Xs + kr * XW + wm * H + 4 * li);
                av[0] = t4[0], av[1] = t4[1], av[2] = t4[2], av[3] = t4[3];
            } else {
                const bp_f32x2 t2 = *reinterpret_cast<const bp_f32x2*>(Xs + kr * XW + wm * H + 2 * li);
                av[0] = t2[0], av[1] = t2[1];
            }
            const bp_f32x2 bv = *reinterpret_cast<const bp_f32x2*>(Ds + kr * DW + wn * 64 + 2 * li);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                bp_mfma(acc[mb][0], av[mb], bv[0]);
                bp_mfma(acc[mb][1], av[mb], bv[1]);
            }
        }
        if (more) stash((s + 1) & 1);
        __syncthreads();
    }
    // block (mb, nb), register reg, lane (li, lh): output row m = wm H + MB i + mb with i = (reg & 3) + 8 (reg >> 2) + 4 lh,
    // output column n = 128 ny + 64 wn + 2 li + nb
    float* dst = a.part + (size_t)blockIdx.x * XW * 4 * H;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const int m = wm * H + MB * i + mb;
            const int n = DW * ny + 64 * wn + 2 * li;
            *reinterpret_cast<bp_f32x2*>(dst + (size_t)m * 4 * H + n) = bp_f32x2{ acc[mb][0][reg], acc[mb][1][reg] };
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same product as EXACT bf16 split products (split != 0; the arithmetic of the rollout's gate product):
// every fp32 operand x = x1 + x2 + x3 (three bf16 terms, round-to-nearest-even, exact residuals), all nine cross products on
// v_mfma_f32_32x32x16_bf16 (a bf16 x bf16 product is exact in fp32), fp32 accumulation.  Here BOTH operands are activations, and
// the matrix instruction wants 8 consecutive k = rows q per lane at a fixed column — the transposed access of row-major data:
//   * a thread stages 8 consecutive rows of ONE column (8 four-byte loads, coalesced across the lanes: consecutive columns),
//     splits them in registers (ps_split_frag: 36 vector instructions per 8 values) and writes three 16-byte fragments — each
//     element of X and D is split ONCE per workgroup, not once per wave that multiplies with it;
//   * LDS holds the planes in fragment order, [plane][k half][column] x 16 bytes: a wave's A / B fragment is one conflict-free
//     ds_read_b128, 18 of them (4 + 2 fragments x 3 planes) per 72 MFMAs of a 16-row step.
// Same grid, tile (2H x 128 per workgroup, H x 64 per wave), K slices and reduction as the fp32 form above.  Issue bound: 9 x 32
// cycles per 32 x 32 x 16 block against 8 x 64 on the fp32 instruction (1.78 x), minus the split that does not hide under the MFMAs.
// ---------------------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256, 2) void lstm_wgrad_split_kernel(const WGradArgs a)
{
    constexpr int KT = 16, XW = 2 * H, DW = 128, MB = H / 32, NB = 2, NT = 256;
    constexpr int NGX = 2 * XW / NT;                             // 8-row groups of X per thread and stage (2 at H = 128, 1 at H = 64)
    constexpr int SQ = 3 * 2 * (XW + DW);                        // 16-byte fragments per stage: [plane][k half][column]
    typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
    IC3_DYNAMIC_LDS(float, smem);
    ps_u32x4* const frag = reinterpret_cast<ps_u32x4*>(smem);    // stage b at frag + b * SQ: X planes, then D planes
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = w & 1, wn = w >> 1;
    const int ny = blockIdx.y;
    const long long q0 = (long long)blockIdx.x * a.rows_per_wg;
    long long nq = a.Q - q0;
    if (nq > a.rows_per_wg) nq = a.rows_per_wg;
    if (nq < 0) nq = 0;
    const __amdgpu_buffer_rsrc_t ri = bp_rsrc(a.inp + q0 * a.ldi, nq > 0 ? ((nq - 1) * a.ldi + H) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rh = bp_rsrc(a.h + q0 * H, nq * H * 4);
    const __amdgpu_buffer_rsrc_t rd = bp_rsrc(a.dg + q0 * 4 * H + ny * DW, nq > 0 ? ((nq - 1) * 4 * H + DW) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rl = bp_rsrc(a.row_live ? a.row_live + q0 : a.h, a.row_live ? nq * 4 : 0);
    const int nstages = (int)((nq + KT - 1) / KT);
    // X group g = tid + i NT: column g % XW, rows 8 (g / XW) .. + 7 of the stage; the D group: column tid % DW, k half tid / DW.
    // Lane part of every address in ONE VGPR (column + the group's k half), the stage / row part on the scalar ALU — an soffset
    // that depends on a VGPR, however uniform, costs a waterfall loop per load.  Which side of [inp | h] a wave stages is
    // wave-uniform (64 consecutive columns): a scalar branch.
    const int dcol = tid % DW, dkg = tid / DW;
    int xoff[NGX], loff[NGX];
    bool x_is_h[NGX];
#pragma unroll
    for (int i = 0; i < NGX; ++i) {
        const int g = tid + i * NT, c = g % XW, kg = g / XW;
        x_is_h[i] = __builtin_amdgcn_readfirstlane((int)(c >= H)) != 0;
        xoff[i] = x_is_h[i] ? ((c - H) + 8 * kg * H) * 4 : (c + 8 * kg * a.ldi) * 4;
        loff[i] = 8 * kg * 4;
    }
    const int doff = (dcol + 8 * dkg * 4 * H) * 4;
    float xv[NGX][8], dv[8];
    auto fetch = [&](int s) {
        const int qb = s * KT;
#pragma unroll
        for (int i = 0; i < NGX; ++i) {
            if (x_is_h[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[i][j] = bp_load1(rh, xoff[i], (qb + j) * H * 4);
                if (a.row_live) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[i][j] *= bp_load1(rl, loff[i], (qb + j) * 4);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[i][j] = bp_load1(ri, xoff[i], (qb + j) * a.ldi * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = bp_load1(rd, doff, (qb + j) * 4 * H * 4);
    };
    auto stash = [&](int b) {
        ps_u32x4* f = frag + b * SQ;
        ps_u32x4 pl[3];
#pragma unroll
        for (int i = 0; i < NGX; ++i) {
            const int g = tid + i * NT, c = g % XW, kg = g / XW;
            ps_split_frag(ps_f32x4{ xv[i][0], xv[i][1], xv[i][2], xv[i][3] }, ps_f32x4{ xv[i][4], xv[i][5], xv[i][6], xv[i][7] }, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) f[(p * 2 + kg) * XW + c] = pl[p];
        }
        ps_split_frag(ps_f32x4{ dv[0], dv[1], dv[2], dv[3] }, ps_f32x4{ dv[4], dv[5], dv[6], dv[7] }, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[3 * 2 * XW + (p * 2 + dkg) * DW + dcol] = pl[p];
    };
    bp_f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.0f;
    // Pipeline: at the top of iteration s the registers hold stage s + 1 (requested a whole iteration ago) and LDS buffer s & 1
    // holds stage s.  The iteration is ONE basic block — the split + stash of stage s + 1 and the 72 products of stage s
    // (independent work: different LDS buffers; the compiler interleaves them), then the loads of stage s + 2 — so that the vector
    // work rides in the issue slots the matrix instructions leave (measured: 6.8 ms with the stash behind a branch and waterfall
    // loops around the loads, 5.8 ms like this, for 3.3 M rows at H = 128; the fp32 form takes 6.4 ms).  Everything is unconditional: stages past the slice read zeros (range check) and
    // stash them into a buffer nobody multiplies.
    fetch(0);
    stash(0);
    fetch(1);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
        const ps_u32x4* fx = frag + (s & 1) * SQ;
        const ps_u32x4* fd = fx + 3 * 2 * XW;
        ps_u32x4 bf[3][NB];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[p][nb] = fd[(p * 2 + lh) * DW + 64 * wn + 32 * nb + li];
        stash((s + 1) & 1);                                      // (in front of the products in program order: the scheduler
                                                                 //  starts the vector work while the first fragments arrive)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            ps_u32x4 af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = fx[(p * 2 + lh) * XW + wm * H + 32 * mb + li];
#pragma unroll
            for (int pb = 0; pb < 3; ++pb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pa = 2; pa >= 0; --pa)                  // (least significant term first)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, af[pa]),
                                                                              __builtin_bit_cast(wg_bf16x8, bf[pb][nb]), acc[mb][nb], 0, 0, 0);
        }
        fetch(s + 2);
        __syncthreads();
    }
    // block (mb, nb), register reg, lane (li, lh): output row m = wm H + 32 mb + (reg & 3) + 8 (reg >> 2) + 4 lh, column
    // n = 128 ny + 64 wn + 32 nb + li
    float* dst = a.part + (size_t)blockIdx.x * XW * 4 * H;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = wm * H + 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                dst[(size_t)m * 4 * H + DW * ny + 64 * wn + 32 * nb + li] = acc[mb][nb][reg];
            }
}

// dW += the K slices' partials, summed in slice order (reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ dW,
                                                           int accumulate)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    dW[i] = accumulate ? dW[i] + s : s;
}

static int bp_cus()
{
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

}  // namespace ic3

// ---- ic3_comm_backward -------------------------------------------------------------------------------------------------------
extern "C" int ic3_comm_backward_partials(int E, int N)
{
    if (E <= 0 || N <= 0 || N > 64) return 0;
    const int ept = 64 / N, tiles = (E + ept - 1) / ept;
    constexpr int cap = 512;
    // at most 512 workgroups (two per CU; 768 / 1024 slots measured: 109 M against 111 M agent-steps/s per PP-hard update), every one
    // the same number of tiles: the launch lasts as long as the workgroup with the most tiles either way, and every workgroup fewer
    // is 2 x H x H x 4 bytes of partial sums less to add to (a chain of 4096 envs of 10 agents: 683 tiles as 342 x 2, not 171 x 2 + 341 x 1)
    const int rounds = (tiles + cap - 1) / cap;
    return (tiles + rounds - 1) / rounds;
}

extern "C" int ic3_comm_backward(const float* dxh, int ldd, const float* h_prev, const int32_t* alive, const int32_t* gate,
                                 const float* c_weight, const float* out_scale, float* dh_out,
                                 float* dcw_partials, int accumulate, int E, int N, int H, int mode_avg, int comm_zero,
                                 ic3_stream stream)
{
    using namespace ic3;
    if (!dxh || !dh_out || E <= 0 || N <= 0) return fail(-22, "ic3_comm_backward: null argument");
    if (ldd < 2 * H || (ldd & 3) || (H & 3)) return fail(-22, "ic3_comm_backward: ldd a multiple of 4, >= 2 * hid_size");
    hipStream_t s = (hipStream_t)stream;
    if (comm_zero) {                                             // no communication: the gate product's share alone
        const long long R = (long long)E * N, n4 = R * (H / 4);
        const int blocks = (int)std::min<long long>((n4 + 255) / 256, 4096);
        hipLaunchKernelGGL(dh_copy_kernel, dim3(blocks), dim3(256), 0, s, dxh, ldd, out_scale, dh_out, R, H / 4);
        IC3_HIP(hipGetLastError());
        return 0;
    }
    if (!h_prev || !c_weight || !dcw_partials) return fail(-22, "ic3_comm_backward: null argument");
    if (H != 64 && H != 128) return fail(-38, "ic3_comm_backward: hid_size 64 / 128");
    if (N > 64) return fail(-38, "ic3_comm_backward: at most 64 agents per env");
    const int ept = 64 / N, tiles = (E + ept - 1) / ept;
    if ((long long)64 * ldd * 4 >= (1ll << 31)) return fail(-22, "ic3_comm_backward: row stride too large");
    const CommBwdArgs a{ dxh, h_prev, alive, gate, c_weight, out_scale, dh_out, dcw_partials, ldd, E, N, ept, tiles,
                         mode_avg, accumulate };
    const int grid = ic3_comm_backward_partials(E, N);
    const size_t lds = ((size_t)64 * (H + 4) + 128) * sizeof(float);
    if (H == 128) {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(comm_bwd_kernel<128>), lds));
        hipLaunchKernelGGL((comm_bwd_kernel<128>), dim3(grid), dim3(256), lds, s, a);
    } else {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(comm_bwd_kernel<64>), lds));
        hipLaunchKernelGGL((comm_bwd_kernel<64>), dim3(grid), dim3(128), lds, s, a);
    }
    IC3_HIP(hipGetLastError());
    return grid;     // rows of dcw_partials written
}

// ---- ic3_lstm_weight_grad ----------------------------------------------------------------------------------------------------
static int wgrad_slices(long long Q, int H)
{
    const int ny = 4 * H / 128;
    int ks = 2 * ic3::bp_cus() / ny;                             // two workgroups per CU
    const long long most = (Q + 15) / 16;
    if (ks > most) ks = (int)most;
    return ks < 1 ? 1 : ks;
}

extern "C" size_t ic3_lstm_weight_grad_scratch_floats(long long Q, int H)
{
    if (Q <= 0 || (H != 64 && H != 128)) return 0;
    return (size_t)wgrad_slices(Q, H) * 2 * H * 4 * H;
}

extern "C" int ic3_lstm_weight_grad(const float* inp, int ldi, const float* h_prev, const float* dgates, const float* row_live,
                                    long long Q, int H, float* dW, int accumulate, int split, float* scratch, ic3_stream stream)
{
    using namespace ic3;
    if (!inp || !h_prev || !dgates || !dW || !scratch || Q <= 0) return fail(-22, "ic3_lstm_weight_grad: null argument");
    if (H != 64 && H != 128) return fail(-38, "ic3_lstm_weight_grad: hid_size 64 / 128");
    if (ldi < H || (ldi & 3)) return fail(-22, "ic3_lstm_weight_grad: ldi a multiple of 4, >= hid_size");
    const int ks = wgrad_slices(Q, H);
    long long per = (Q + ks - 1) / ks;
    per = (per + 15) / 16 * 16;
    if (per * (long long)std::max(ldi, 4 * H) * 4 >= (1ll << 31))
        return fail(-22, "ic3_lstm_weight_grad: a K slice must stay below 2 GB per operand (32-bit buffer offsets)");
    const WGradArgs a{ inp, h_prev, dgates, row_live, scratch, Q, ldi, (int)per };
    hipStream_t s = (hipStream_t)stream;
    if (split) {                                                 // exact bf16 split products (the rollout's arithmetic)
        const size_t lds3 = (size_t)2 * 3 * 2 * (2 * H + 128) * 16;
        if (H == 128) {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_wgrad_split_kernel<128>), lds3));
            hipLaunchKernelGGL((lstm_wgrad_split_kernel<128>), dim3(ks, 4), dim3(256), lds3, s, a);
        } else {
            IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_wgrad_split_kernel<64>), lds3));
            hipLaunchKernelGGL((lstm_wgrad_split_kernel<64>), dim3(ks, 2), dim3(256), lds3, s, a);
        }
        IC3_HIP(hipGetLastError());
        const int n3 = 2 * H * 4 * H;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n3 + 255) / 256), dim3(256), 0, s, scratch, ks, n3, dW, accumulate);
        IC3_HIP(hipGetLastError());
        return ks;
    }
    const size_t lds = (size_t)2 * 16 * (2 * H + 128) * sizeof(float);
    if (H == 128) {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_wgrad_kernel<128>), lds));
        hipLaunchKernelGGL((lstm_wgrad_kernel<128>), dim3(ks, 4), dim3(256), lds, s, a);
    } else {
        IC3_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lstm_wgrad_kernel<64>), lds));
        hipLaunchKernelGGL((lstm_wgrad_kernel<64>), dim3(ks, 2), dim3(256), lds, s, a);
    }
    IC3_HIP(hipGetLastError());
    const int n = 2 * H * 4 * H;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, scratch, ks, n, dW, accumulate);
    IC3_HIP(hipGetLastError());
    return ks;
}

// ---- ic3_bptt_backward -------------------------------------------------------------------------------------------------------
// The loop runs in place on the record (dgates over the gates): whatever could refuse a step is asked BEFORE the first one.
extern "C" int ic3_bptt_backward_supported(const ic3_env* env, int H)
{
    using namespace ic3;
    if (!env || (H != 64 && H != 128) || env->dims.N > 64) return 0;
    EncBwdPlan pl;
    if (env->kind == IC3_ENV_PP) {
        const ic3_pp_cfg& c = env->pp;
        const int W = 2 * c.vision + 1;
        pl = enc_bwd_plan(c.E, env->dims.N, c.N + c.nprey, H, c.dim * c.dim, 2 * W * W);
    } else {
        const ic3_tj_cfg& c = env->tj;
        const int WW = env->dims.window * env->dims.window, hdr = c.vocab_type ? 4 : 2;
        pl = enc_bwd_plan(c.E, c.N, c.N, H, env->dims.grid_h * env->dims.grid_w, hdr + WW);
    }
    return pl.csplit ? 1 : 0;     // (the sparse encoder's backward in its partial-sums form: ic3_env_encode_backward_accumulate)
}

namespace ic3 {
// the second chain's stream + the fork / join events, per device
struct BpttSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static BpttSide* bptt_side()
{
    static BpttSide side[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    BpttSide& sd = side[dev];
    if (!sd.stream) {
        if (hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess)
            return nullptr;
    }
    return &sd;
}
}  // namespace ic3

// envs [0, E1) run on the caller's stream, [E1, E) on the library's second stream (ic3_bptt.two_chains): E1 * N a multiple of 64,
// so that the gate launch's row tiles — and with them its bias partials — do not straddle the border
extern "C" int ic3_bptt_first_chain_envs(int E, int N)
{
    if (E < 128 || N <= 0) return E;
    return (E / 2) & ~63;
}

extern "C" int ic3_bptt_backward(ic3_env* env, const ic3_bptt* b, ic3_stream stream)
{
    using namespace ic3;
    if (!env || !b) return fail(-22, "ic3_bptt_backward: null argument");
    if (b->struct_size != sizeof(ic3_bptt))
        return fail(-22, "ic3_bptt_backward: ic3_bptt has " + std::to_string(b->struct_size) + " bytes, this library's has " +
                             std::to_string(sizeof(ic3_bptt)) + " (header / library version mismatch)");
    const int T = b->T, E = b->E, N = b->N, H = b->H;
    if (T <= 0 || E <= 0 || N <= 0 || E != env->dims.E || N != env->dims.N)
        return fail(-22, "ic3_bptt_backward: T, E, N must be positive and E, N the handle's");
    if (!ic3_bptt_backward_supported(env, H)) return fail(-38, "ic3_bptt_backward: hid_size 64 / 128, <= 64 agents, a grid whose encoder backward runs in its partial-sums form");
    if (b->OT < 1 || b->OT > 16) return fail(-22, "ic3_bptt_backward: 1 <= OT <= 16");
    if (!b->gates || !b->hs || !b->cs || !b->dhead || !b->snaps || !b->lstm_wp3_bwd || !b->w_heads || !b->dh || !b->dc || !b->dxh ||
        !b->dbias_partials || !b->enc_work)
        return fail(-22, "ic3_bptt_backward: null argument");
    if (!b->comm_zero && (!b->c_weight || !b->dcw_partials)) return fail(-22, "ic3_bptt_backward: C.weight and its partials");
    if (b->dxh_step && (b->dxh_step < (long long)E * N * 2 * H || ic3_env_encode_backward_window_work(env, H) <= 0))
        return fail(-22, "ic3_bptt_backward: dxh_step >= E * N * 2 * hid_size, on a configuration with ic3_env_encode_backward_window");
    const long long R = (long long)E * N;
    hipStream_t s = (hipStream_t)stream;
    // Two chains: the steps of envs [0, E1) and [E1, E) are independent until the weight gradients are summed, so with
    // two_chains their launches go to two streams and the GPU fills the ragged last round of one chain's launch (1280 row tiles
    // on 512 workgroup slots = 2.5 rounds at PP-hard E = 8192) with the other chain's workgroups.  Needs what ties the chains
    // together to be per step or behind the loop: the ring of input gradients (the encoder's stage 1 behind the loop).
    const int E1 = (b->two_chains && b->dxh_step) ? ic3_bptt_first_chain_envs(E, N) : E;
    const int nch = E1 < E ? 2 : 1;
    BpttSide* side = nch == 2 ? bptt_side() : nullptr;
    if (nch == 2 && !side) return fail(-12, "ic3_bptt_backward: no second stream");
    if (nch == 2) {
        IC3_HIP(hipEventRecord(side->fork, s));
        IC3_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
    }
    int enc_first = b->enc_first;
    for (int t = T - 1; t >= 0; --t) {
        for (int ch = 0; ch < nch; ++ch) {
            const int e0 = ch ? E1 : 0, Ec = ch ? E - E1 : E1;
            const size_t r0 = (size_t)e0 * N, Rc = (size_t)Ec * N;
            hipStream_t sc = ch ? side->stream : s;
            ic3_stream scv = (ic3_stream)sc;
            float* dh = b->dh + r0 * H;
            float* dc = b->dc + r0 * H;
            // trainer.py:56-60: (h_t, c_t) were handed on detached — the gate launch reads zeros for dL/d(h_t, c_t) (null inputs: an
            // empty descriptor instead of two memsets and their reads)
            const bool detached = b->detach_gap > 0 && (t + 1) % b->detach_gap == 0;
            float* g = b->gates + ((size_t)t * R + r0) * 4 * H;
            float* dxh = b->dxh + (size_t)t * (size_t)b->dxh_step + r0 * 2 * H;
            if (b->gate_events && ch == 0) IC3_HIP(hipEventRecord((hipEvent_t)b->gate_events[2 * t], sc));
            int rc = ic3_lstm_gates_backward_given(g, nullptr, 0, nullptr, b->lstm_wp3_bwd, b->cs + ((size_t)t * R + r0) * H,
                                                   detached ? nullptr : dh, detached ? nullptr : dc, g,
                                                   dc, b->dbias_partials + (r0 / 64) * 4 * H, 1, dxh,
                                                   b->row_live ? b->row_live + (size_t)t * R + r0 : nullptr,
                                                   b->row_keep ? b->row_keep + (size_t)t * R + r0 : nullptr,
                                                   b->dhead + ((size_t)t * R + r0) * b->OT, b->w_heads, b->OT, (int)Rc, H, scv);
            if (rc < 0) return rc;
            if (b->gate_events && ch == 0) IC3_HIP(hipEventRecord((hipEvent_t)b->gate_events[2 * t + 1], sc));
            const float* out_scale = (b->row_keep && t > 0) ? b->row_keep + (size_t)(t - 1) * R + r0 : nullptr;
            float* dcw = b->dcw_partials;
            if (dcw && ch) dcw += (size_t)ic3_comm_backward_partials(E1, N) * H * H;
            rc = ic3_comm_backward(dxh, 2 * H, b->hs + ((size_t)t * R + r0) * H, (b->alive && b->alive[t]) ? b->alive[t] + r0 : nullptr,
                                   (b->gate && b->gate[t]) ? b->gate[t] + r0 : nullptr, b->c_weight, out_scale, dh, dcw, 1, Ec, N, H,
                                   b->mode_avg, b->comm_zero, scv);
            if (rc < 0) return rc;
        }
        if (b->dxh_step) continue;                               // (the encoder's first stage: once, behind the loop)
        int rc = ic3_env_encode_backward_accumulate(env, b->snaps + (size_t)t * b->snap_words, b->dxh, 2 * H, H, b->enc_work,
                                                    enc_first, stream);
        if (rc < 0) return rc;
        enc_first = 0;
    }
    if (nch == 2) {
        IC3_HIP(hipEventRecord(side->join, side->stream));
        IC3_HIP(hipStreamWaitEvent(s, side->join, 0));
    }
    if (b->dxh_step)
        return ic3_env_encode_backward_window(env, b->snaps, b->snap_words, T, b->dxh, 2 * H, b->dxh_step, H, b->enc_work, enc_first,
                                              stream);
    return 0;
}
