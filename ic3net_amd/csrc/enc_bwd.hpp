// enc_bwd.hpp — backward of the sparse observation encoder (ic3_env_encode_backward).
//
// forward (pp_encode_kernel / tj_encode_kernel):  enc[row] = bias + sum_cells Wt[col(cell, id(pos[row], cell))]
//                                                           + sum_slots cnt(row, slot) * Wt[col(slot)]
// where id(pos, cell) depends only on the agent's grid position, and the "slots" are the few obs channels whose
// column is the same for every row (per-cell class counts, the TJ header scalars).  Hence for g = dL/d enc:
//   dWt[col(cell, id(pos, cell))] += P[pos],   P[pos] = sum_{rows at pos} g[row]        (position sums)
//   dWt[col(slot)]                 = D[slot],   D[slot] = sum_rows cnt(row, slot) * g[row]
//   dbias                          = sum_rows g[row]                                     (slot index nslots)
// Stage 1 (accumulate, one workgroup per chunk of envs): P by float atomics (R*H of them instead of R*cells*H),
// D in LDS accumulators with a fixed (slot, h) -> thread ownership, flushed once per workgroup as a partial.
// Stage 2 (expand): scatters P through the id map into the zeroed dWt and reduces the partials.
// This replaces the dense obs^T x g GEMM of the nn.Linear backward (2*R*obs_dim*H flops) — trainer.py:128-225 path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ps_common.hpp"

namespace ic3 {

typedef float encb_f32x4 __attribute__((ext_vector_type(4)));
constexpr int ENCB_SPLIT = 16;   // workgroup partials summed per thread in the expand stage

// One env's contribution.  gl: LDS [rows*H] staging of g; Dl: LDS [(nslots+1)*H] accumulators (owned per idx).
template <class Cnt, class Pos>
__device__ __forceinline__ void enc_bwd_accumulate(const float* __restrict__ g, int ldg, size_t row0, int rows, int H,
                                                   float* gl, float* Dl, int nslots, float* __restrict__ P, Cnt cnt,
                                                   Pos pos)
{
    const int H4 = H >> 2;
    for (int idx = threadIdx.x; idx < rows * H4; idx += blockDim.x) {
        const int a = idx / H4, c4 = idx - a * H4;
        const encb_f32x4 v = reinterpret_cast<const encb_f32x4*>(g + (row0 + a) * (size_t)ldg)[c4];
        reinterpret_cast<encb_f32x4*>(gl)[idx] = v;
        const int p = pos(a);
        if (p >= 0) {      // (round 3: PP-hard E = 8192 takes 212 us per call with these atomics, 81 us without: see enc_bwd_rows below)
            float* dst = P + (size_t)p * H + 4 * c4;
            if (v.x != 0.f) atomicAdd(dst + 0, v.x);
            if (v.y != 0.f) atomicAdd(dst + 1, v.y);
            if (v.z != 0.f) atomicAdd(dst + 2, v.z);
            if (v.w != 0.f) atomicAdd(dst + 3, v.w);
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < (nslots + 1) * H; idx += blockDim.x) {
        const int s = idx / H, h = idx - s * H;
        float acc = 0.f;
        if (s == nslots) {
            for (int a = 0; a < rows; ++a) acc += gl[a * H + h];
        } else {
            for (int a = 0; a < rows; ++a) {
                const float c = cnt(a, s);
                if (c != 0.f) acc += c * gl[a * H + h];
            }
        }
        Dl[idx] += acc;
    }
    __syncthreads();  // the next env overwrites gl and the window table
}

// ---------------------------------------------------------------------------------------------------------------------
// Stage 1, second form (round 3): rows in parallel instead of one env at a time.
// The form above walks a workgroup's envs one after the other (three barriers per env, 1280 floats of work between
// them at PP-hard) and adds every element of g to P with a global float atomic: 212 us per call at PP-hard E = 8192 for
// 42 MB of input (131 us of it the atomics).  Here a thread owns (row lane, float4 column chunk) of a COLUMN SLICE of
// Hc = H / csplit columns (blockIdx.y) and strides over the rows of its row group (blockIdx.x); P and D of the slice
// are accumulated in LDS (ds_add_f32: npos * Hc + (nslots + 1) * Hc floats, zeroed at the start) and written out once
// per workgroup as partials Ppart[rowgroup][pos][H] / Dpart[rowgroup][slot][H] that the expand stage sums.  A row
// contributes its g to P[pos(row)], to the bias row, and wgt * g to every slot `row_fn` reports through `emit` — the
// entities inside the row's window, found by a loop over the env's entities instead of a per-cell count table.
// Chosen by enc_bwd_plan() when the slice fits in LDS (npos * 16 bytes <= the budget); otherwise the first form runs.
struct EncBwdPlan {
    int csplit = 0;   // column slices (0 = does not fit: use the per-env form)
    int nrg = 0;      // row groups = partials
    int lds = 0;      // bytes
};
inline EncBwdPlan enc_bwd_plan(int E, int rows_env, int total, int H, int npos, int nslots)
{
    EncBwdPlan p;
    if (H > 256 || (H & 3) || total > 1024 || rows_env > total) return p;
    const int stage_words = 2 * 1024 + 3 * 256 + 2048 + 4;       // ent + order (ENCB_STAGE_WORDS each) + cnt / start / cursor + hits
    for (int cs = 1; cs <= H / 4; cs <<= 1) {
        if (H % (4 * cs)) break;
        const int Hc = H / cs;
        if (Hc > 256 * 4 || 256 % (Hc / 4)) continue;
        const size_t lds = ((size_t)npos * Hc + (size_t)(nslots + 1) * Hc + stage_words) * sizeof(float);
        if (lds <= 80 * 1024) {   // two workgroups per CU
            p.csplit = cs;
            p.lds = (int)lds;
            break;
        }
    }
    if (!p.csplit) return p;
    const int RL = 256 / (H / p.csplit / 4);                     // row lanes of a workgroup
    int nrg = 512 / p.csplit;                                    // ~two workgroups per CU in all
    const long long R = (long long)E * rows_env;
    const int most = (int)((R + 4 * RL - 1) / (4 * RL));         // at least four passes of rows per workgroup
    if (nrg > most) nrg = most;
    if (nrg < 1) nrg = 1;
    const int per_wg = (E + nrg - 1) / nrg;                      // whole envs per workgroup
    p.nrg = (E + per_wg - 1) / per_wg;
    return p;
}

__device__ __forceinline__ void lds_add4(float* dst, encb_f32x4 v)
{
    if (v.x != 0.f) atomicAdd(dst + 0, v.x);
    if (v.y != 0.f) atomicAdd(dst + 1, v.y);
    if (v.z != 0.f) atomicAdd(dst + 2, v.z);
    if (v.w != 0.f) atomicAdd(dst + 3, v.w);
}

// A workgroup owns a contiguous range of ENVS (gridDim.x ranges) and walks it in batches of <= ENCB_STAGE_WORDS / total
// envs.  Per batch:
//   1. the entity positions are staged in LDS as packed (row | col << 16) words — `stage(i)` = the word of entity i of the
//      flat [E][total] position arrays — so that the per-row loop over the env's entities reads LDS;
//   2. the batch's rows are counting-sorted by  bin = pos % RL  (RL = row lanes of the workgroup; rows without a position
//      are spread round-robin): row lane b then owns every row whose position falls in bin b, so P[pos] += g[row] is a
//      plain 16-byte LDS read-modify-write — no two lanes ever touch the same (pos, column).  (LDS float atomics measured
//      ~2.6 cycles per LANE on gfx950 — 45 us per call for P alone at PP-hard E = 8192 — and an address every row hits,
//      the observer's own window centre, serialises completely: 570 us.)
//   3. what EVERY row adds to (its own centre cell, the TJ header scalars) accumulates in registers: row_fn reports it
//      through reg(k, weight), k < ENCB_REGS a literal; `reg_slot(k)` names the slot register k is flushed into at the end.
//   4. the OTHER entities inside a row's window are rare (0.2 per row at PP-hard): all (row, entity) pairs of the batch are
//      tested ONCE, ENCB_HITS of them per round over the whole workgroup — pair_fn(ent, a, p, row) = the slot entity p
//      feeds in row a's window, or -1 — the hits are listed in LDS and then added with LDS atomics by the row lanes.
//      (A per-row loop over the env's entities inside the row pass cost 40 us per call: a dependent LDS read and a
//      divergent branch per entity in front of every row.)
// RowFn: int pos = row_fn(row, ent, a, reg): `ent` = the staged words of the row's env, `a` = the row's index inside it;
// pos < 0 = no position term.  KeyFn: the same pos from (ent, a, row) alone (the sort key).
constexpr int ENCB_STAGE_WORDS = 1024;   // staged positions per batch (4 KB); also bounds the rows of a batch
constexpr int ENCB_REGS = 5;
constexpr int ENCB_HITS = 2048;          // (row, entity) pairs tested per round = capacity of the hit list (8 KB)
template <class StageFn, class KeyFn, class RowFn, class PairFn, class RegSlotFn>
__device__ __forceinline__ void enc_bwd_rows(const float* __restrict__ g, int ldg, int E, int rows_env, int total, int H,
                                             int Hc, int npos, int nslots, float* __restrict__ Ppart,
                                             float* __restrict__ Dpart, float* sm, StageFn stage, KeyFn key_fn, RowFn row_fn,
                                             PairFn pair_fn, RegSlotFn reg_slot, int accumulate = 0)
{
    const int C4 = Hc >> 2, RL = 256 / C4;
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
    const int col0 = blockIdx.y * Hc;
    encb_f32x4* Pl4 = reinterpret_cast<encb_f32x4*>(sm);
    float* Dl = sm + (size_t)npos * Hc;
    int32_t* ent = reinterpret_cast<int32_t*>(Dl + (size_t)(nslots + 1) * Hc);   // [ENCB_STAGE_WORDS]
    int32_t* order = ent + ENCB_STAGE_WORDS;                                      // [ENCB_STAGE_WORDS] rows of the batch by bin
    int32_t* cnt = order + ENCB_STAGE_WORDS;                                      // [256] rows per bin
    int32_t* start = cnt + 256;                                                   // [256] first slot of a bin in `order`
    int32_t* cursor = start + 256;                                                // [256]
    int32_t* hits = cursor + 256;                                                 // [ENCB_HITS] local row | slot << 16
    int32_t* nhit = hits + ENCB_HITS;                                             // [4]
    const int nz4 = (npos + nslots + 1) * C4;
    for (int i = threadIdx.x; i < nz4; i += 256) Pl4[i] = encb_f32x4{ 0.f, 0.f, 0.f, 0.f };
    const int per_wg = (E + (int)gridDim.x - 1) / (int)gridDim.x;
    const int e_begin = blockIdx.x * per_wg, e_end = min(E, e_begin + per_wg);
    const int batch = max(1, ENCB_STAGE_WORDS / total);
    encb_f32x4 db = { 0.f, 0.f, 0.f, 0.f }, racc[ENCB_REGS];
#pragma unroll
    for (int k = 0; k < ENCB_REGS; ++k) racc[k] = encb_f32x4{ 0.f, 0.f, 0.f, 0.f };
    for (int eb = e_begin; eb < e_end; eb += batch) {
        const int nb = min(batch, e_end - eb), nrows = nb * rows_env;
        __syncthreads();                                         // the zero fill / the previous batch's readers
        for (int i = threadIdx.x; i < nb * total; i += 256) ent[i] = stage((size_t)eb * total + i);
        cnt[threadIdx.x] = 0;
        __syncthreads();
        auto bin_of = [&](int lr) {
            const int el = lr / rows_env, a = lr - el * rows_env;
            const int pos = key_fn(ent + el * total, a, (size_t)(eb + el) * rows_env + a);
            return (pos < 0 ? lr : pos) % RL;
        };
        for (int lr = threadIdx.x; lr < nrows; lr += 256) atomicAdd(&cnt[bin_of(lr)], 1);
        __syncthreads();
        if ((int)threadIdx.x < RL) {
            int s0 = 0;
            for (int j = 0; j < (int)threadIdx.x; ++j) s0 += cnt[j];
            start[threadIdx.x] = s0;
            cursor[threadIdx.x] = s0;
        }
        __syncthreads();
        for (int lr = threadIdx.x; lr < nrows; lr += 256) order[atomicAdd(&cursor[bin_of(lr)], 1)] = lr;
        __syncthreads();
        const int i1 = start[rl] + cnt[rl];
        for (int i = start[rl]; i < i1; ++i) {
            const int lr = order[i];
            const int el = lr / rows_env, a = lr - el * rows_env;
            const size_t row = (size_t)(eb + el) * rows_env + a;
            const encb_f32x4 v = *reinterpret_cast<const encb_f32x4*>(g + row * ldg + col0 + 4 * c4);
            db += v;
            const int pos = row_fn(row, ent + el * total, a, [&](int k, float wgt) { racc[k] += wgt * v; });
            if (pos >= 0) Pl4[pos * C4 + c4] += v;     // this lane owns bin pos % RL
        }
        // the other entities inside the windows
        const int npairs = nrows * total;
        for (int p0 = 0; p0 < npairs; p0 += ENCB_HITS) {
            if (threadIdx.x == 0) nhit[0] = 0;
            __syncthreads();
            for (int idx = p0 + threadIdx.x; idx < min(npairs, p0 + ENCB_HITS); idx += 256) {
                const int lr = idx / total, p = idx - lr * total;
                const int el = lr / rows_env, a = lr - el * rows_env;
                const int slot = pair_fn(ent + el * total, a, p, (size_t)(eb + el) * rows_env + a);
                if (slot >= 0) hits[atomicAdd(&nhit[0], 1)] = lr | (slot << 16);
            }
            __syncthreads();
            const int nh = nhit[0];
            for (int i = rl; i < nh; i += RL) {
                const int lr = hits[i] & 0xffff, slot = hits[i] >> 16;
                const int el = lr / rows_env, a = lr - el * rows_env;
                const size_t row = (size_t)(eb + el) * rows_env + a;
                lds_add4(Dl + slot * Hc + 4 * c4, *reinterpret_cast<const encb_f32x4*>(g + row * ldg + col0 + 4 * c4));
            }
            __syncthreads();
        }
    }
    __syncthreads();                                             // (also orders the zero fill for an empty env range)
    lds_add4(Dl + nslots * Hc + 4 * c4, db);
#pragma unroll
    for (int k = 0; k < ENCB_REGS; ++k) {
        const int slot = reg_slot(k);
        if (slot >= 0) lds_add4(Dl + slot * Hc + 4 * c4, racc[k]);
    }
    __syncthreads();
    float* Pg = Ppart + (size_t)blockIdx.x * npos * H + col0;
    for (int i = threadIdx.x; i < npos * C4; i += 256) {
        const int pos = i / C4, k = i - pos * C4;
        // accumulate (ic3_env_encode_backward_accumulate): the workgroup's partial of the previous calls + this call's — the
        // expand stage is linear in the partials, so a whole episode's states need it ONCE
        encb_f32x4* dst = reinterpret_cast<encb_f32x4*>(Pg + (size_t)pos * H + 4 * k);
        *dst = accumulate ? *dst + Pl4[i] : Pl4[i];
    }
    float* Dg = Dpart + (size_t)blockIdx.x * (nslots + 1) * H + col0;
    for (int i = threadIdx.x; i < (nslots + 1) * C4; i += 256) {
        const int s = i / C4, k = i - s * C4;
        encb_f32x4* dst = reinterpret_cast<encb_f32x4*>(Dg + (size_t)s * H + 4 * k);
        *dst = accumulate ? *dst + reinterpret_cast<const encb_f32x4*>(Dl)[i] : reinterpret_cast<const encb_f32x4*>(Dl)[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stage 1, third form (round 6): a whole WINDOW of T recorded states in one launch, on the matrix cores.
//   out[m][h] = sum over the Q = T * R rows q of  A[m][q] * g[q][h],     m < npos:  A = [pos(q) == m]   (a one-hot column)
//                                                                         m = npos + slot:  A = the slot's weight in row q
//                                                                         m = npos + nslots:  A = 1      (the bias)
// is a product with K = Q.  The one-hot entries are exact in bf16; g is split exactly into three bf16 terms
// (ps_split_frag), so a 32-position block costs 3 v_mfma_f32_32x32x16_bf16 per 16 rows; the slot weights (counts at PP, arbitrary
// header scalars at TJ) sit in an LDS table [slot][row of the batch] that the workgroup fills per batch of ENCW_RB rows — the
// row itself, then every (row, other entity) pair that falls into the row's window, the entities' positions staged in LDS — and
// are split the same way: 9 products for their block (3 where the weights are small counts: Spec::slots_exact_bf16).  The one-hot
// fragments are the same for every wave of the workgroup (the waves differ in their columns of g) and all but empty: they live
// in LDS (8 KB per position block), a row's owner sets its one entry and clears it behind the products, the product loop reads a
// fragment with one ds_read_b128.  (Made in registers inside the loop — 3 packed 16-bit instructions per pair of rows, 156 per
// K step at 13 blocks — the loop was issue-bound at 2.6 x its matrix time; made per batch by compares, the batch preparation
// took as long as the batch's products.)  A wave owns 32 columns of g (its B fragment: 8 rows x 1 column per lane,
// straight from global memory, one K step ahead) and MBP position blocks + 1 slot block of `out` in registers for the WHOLE row
// range of its workgroup (gridDim.z slices the blocks: slice z holds position blocks [z MBP, (z + 1) MBP) and slot block z); the
// partials are written once per launch in the layout of the second form (Ppart[row group][pos][H], Dpart[row group][slot][H])
// for the same expand stage.
// Against the second form per step (84 us per state at PP-hard E = 8192: an LDS counting sort per batch, one partial read-modify-
// write per launch): 80 states in one launch, no sort, no LDS accumulators.  Needs the rows' g of ALL the window's steps at
// once: ic3_bptt.dxh_step (a ring of per-step input gradients instead of one buffer).
// Spec (per env kind; all methods const, device):  total, rows_env;  word(st, e, i) = packed (row | col << 16) of entity i;
// live(st, row);  pos(word);  self(st, row, a, word, emit(slot, weight));  pair(word_a, word_p, a, p) = slot or -1.
constexpr int ENCW_RB = 128;             // rows per batch of the slot table (8 K steps)
constexpr int ENCW_LDW = ENCW_RB + 4;    // floats per slot row of the table
struct EncWinPlan {
    int MBP = 0;       // position blocks per wave (template argument: 3, 7 or 13); 0 = not supported
    int nsl = 0;       // slices of the M dimension (gridDim.z)
    int ncs = 0;       // column slices of <= 128 columns (gridDim.y)
    int nw = 0;        // waves per workgroup = columns of a slice / 32
    int nrg = 0;       // row groups (gridDim.x) = partials
    int nstage = 0;    // staged entity words per batch
    int lds = 0;       // bytes
    int PB = 0, SB = 0;
};
inline EncWinPlan enc_win_plan(long long R, int rows_env, int total, int H, int npos, int nslots, int cus)
{
    EncWinPlan p;
    if (H < 32 || (H & 31) || R <= 0 || npos <= 0 || npos >= 0xffff || rows_env <= 0 || total < rows_env) return p;
    p.PB = (npos + 31) / 32;
    p.SB = (nslots + 1 + 31) / 32;
    const int mbp = p.PB <= 3 ? 3 : (p.PB <= 7 ? 7 : 13);
    p.nsl = (p.PB + mbp - 1) / mbp;
    if (p.nsl < p.SB) p.nsl = p.SB;
    p.ncs = (H + 127) / 128;
    if (H % p.ncs || (H / p.ncs) % 32) return p;
    p.nw = H / p.ncs / 32;
    p.nstage = (ENCW_RB / rows_env + 2) * total;
    p.lds = 32 * ENCW_LDW * 4 + (ENCW_RB / 16) * mbp * 64 * 16 + p.nstage * 4;
    if (p.lds > 150 * 1024 || p.nstage > 4 * 64 * p.nw) return p;
    const int wg_per_cu = (mbp == 13 ? 1 : (mbp == 7 ? 2 : 4)) * (4 / p.nw > 0 ? 4 / p.nw : 1);
    long long nrg = (long long)cus * wg_per_cu / ((long long)p.ncs * p.nsl);
    const long long nbat = (R + ENCW_RB - 1) / ENCW_RB;          // (of ONE state: the partial count does not depend on T)
    if (nrg > nbat) nrg = nbat;
    if (nrg < 1) nrg = 1;
    p.nrg = (int)nrg;
    p.MBP = mbp;
    return p;
}
struct EncWinArgs {
    const int32_t* snaps;      // state of step t at snaps + t * snap_words
    long long snap_words;
    const float* g;            // g of (t, row) at g + t * g_step + row * ldg
    long long g_step;
    int ldg, T, E, R, H, npos, nslots, PB, SB, accumulate, nstage;
    int bat_per_wg;            // batches of ENCW_RB rows per row group
    float* Ppart;
    float* Dpart;
};

// x / d for x < 2^31 through m = floor((2^32 - 1) / d): the estimate is the quotient or one below it
__host__ __device__ inline unsigned encw_magic(int d) { return (unsigned)(0xffffffffull / (unsigned)d); }
__device__ __forceinline__ int encw_div(int x, int d, unsigned m)
{
    int qd = (int)__umulhi((unsigned)x, m);
    if (x - qd * d >= d) ++qd;
    return qd;
}

template <int MBP, class Spec>
__device__ __forceinline__ void enc_bwd_window(const EncWinArgs& a, const Spec& sp, unsigned char* sm)
{
    typedef __bf16 encw_bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int NKS = ENCW_RB / 16;
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, w = tid >> 6, nw = NT >> 6, li = lane & 31, lh = lane >> 5;
    float* Wt = reinterpret_cast<float*>(sm);                                          // [32][ENCW_LDW]: this slice's slot block
    ps_u32x4* afr = reinterpret_cast<ps_u32x4*>(Wt + 32 * ENCW_LDW);                   // [NKS][MBP][64] one-hot A fragments
    unsigned* ent = reinterpret_cast<unsigned*>(afr + NKS * MBP * 64);                 // [nstage] entity words of the batch's envs
    const int gb0 = blockIdx.z * MBP;                                                  // first position block of this M slice
    const bool has_pos = gb0 < a.PB, has_slots = (int)blockIdx.z < a.SB;               // (uniform)
    const int s0 = blockIdx.z * 32;                                                    // first slot of this slice's slot block
    const int col = blockIdx.y * (32 * nw) + 32 * w + li;
    const int Q = a.T * a.R;
    const int qb = (int)min((long long)blockIdx.x * a.bat_per_wg * ENCW_RB, (long long)Q);
    const int qe = (int)min((long long)qb + (long long)a.bat_per_wg * ENCW_RB, (long long)Q);
    const unsigned m_rows = encw_magic(sp.rows_env), m_E = encw_magic(a.E);
    ps_f32x16 accp[MBP], accs;
#pragma unroll
    for (int i = 0; i < 16; ++i) accs[i] = 0.f;
#pragma unroll
    for (int b = 0; b < MBP; ++b) accp[b] = accs;
    // The lane's 8 rows of a K step, q = kq + 8 lh + j, column `col`: buffer loads off a descriptor whose base is the K step's
    // first row and whose size ends at the range's last row (rows past it read 0: no branch, no select) — scalar bookkeeping
    // (kq, kt, kr), loop-invariant lane offsets.  A K step that straddles two states whose rows are not contiguous
    // (wrap != 0; the ring of ic3_bptt is contiguous) reads the second state's rows off a second descriptor.
    const long long wrap = a.g_step - (long long)a.R * a.ldg;
    int kq = qb, kt = qb / a.R, kr = qb - kt * a.R;              // (uniform)
    int voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) voff[j] = ((8 * lh + j) * a.ldg + col) * 4;
    auto fetch = [&](float (&v)[8]) {
        const int nv = min(16, qe - kq);                          // rows of this K step inside the range (<= 0: none)
        const float* base = a.g + (size_t)kt * a.g_step + (size_t)kr * a.ldg;
        const bool cross = wrap != 0 && kr + 16 > a.R;            // (uniform)
        const int nA = cross ? min(nv, a.R - kr) : nv;
        const __amdgpu_buffer_rsrc_t rA = make_rsrc(base, nA > 0 ? (uint32_t)nA * a.ldg * 4u : 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf_load_b32(rA, voff[j], 0);
        if (cross && nv > nA) {                                   // rows nA.. of the K step: the next state's first rows
            const float* b2 = a.g + (size_t)(kt + 1) * a.g_step - (size_t)nA * a.ldg;
            const __amdgpu_buffer_rsrc_t rB = make_rsrc(b2, (uint32_t)nv * a.ldg * 4u);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = buf_load_b32(rB, voff[j], 0);
                if (8 * lh + j >= nA) v[j] = x;
            }
        }
        kq += 16;
        kr += 16;
        while (kr >= a.R) { kr -= a.R; ++kt; }
    };
    // the entity words of a batch's envs (flat env fe = q / rows_env = t E + e), one batch ahead in registers
    unsigned ew[4];                                                // (nstage <= 4 * NT: enc_win_plan)
    auto stage_fetch = [&](int q0) {
        const int fe0 = encw_div(q0, sp.rows_env, m_rows);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * NT;
            unsigned v = 0u;
            if (idx < a.nstage && q0 < qe) {
                const int de = idx / sp.total, i = idx - de * sp.total, fe = fe0 + de;
                const int t = encw_div(fe, a.E, m_E), e = fe - t * a.E;
                if (t < a.T) v = sp.word(a.snaps + (size_t)t * a.snap_words, e, i);
            }
            ew[k] = v;
        }
    };
    float g0[8], g1[8], g2[8], g3[8];                              // the next FOUR K steps' rows: 8 KB per wave in flight
    fetch(g0);
    fetch(g1);
    fetch(g2);
    fetch(g3);
    stage_fetch(qb);
    // The one-hot fragments: LDS holds the batch's NKS x MBP of them, all zero but for ONE bf16 1.0 per row (at block pos / 32,
    // lane pos % 32 + 32 (k / 8), element k % 8 of the row's K step) — the thread that owns a row sets it and clears it again
    // behind the products, so the table is zeroed once per launch and a batch costs two 2-byte stores per row.
    for (int i = tid; i < NKS * MBP * 64; i += NT) afr[i] = ps_u32x4{ 0u, 0u, 0u, 0u };
    int mark[ENCW_RB / 64];                                        // byte offset of this thread's mark per row it owns, or -1
#pragma unroll
    for (int k = 0; k < ENCW_RB / 64; ++k) mark[k] = -1;
    unsigned short* const afh = reinterpret_cast<unsigned short*>(afr);
    for (int q0 = qb; q0 < qe; q0 += ENCW_RB) {
        __syncthreads();                                          // the previous batch's readers
#pragma unroll
        for (int k = 0; k < ENCW_RB / 64; ++k)
            if (mark[k] >= 0) afh[mark[k]] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + k * NT < a.nstage) ent[tid + k * NT] = ew[k];
        if (has_slots)
            for (int i = tid; i < 32 * (ENCW_LDW / 4); i += NT) reinterpret_cast<ps_f32x4*>(Wt)[i] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
        __syncthreads();
        stage_fetch(q0 + ENCW_RB);                                // the next batch's words: in flight under this batch's products
        const int fe0 = encw_div(q0, sp.rows_env, m_rows);
#pragma unroll
        for (int k = 0; k < ENCW_RB / 64; ++k) {                  // rows lr = tid + k NT (NT >= 64)
            const int lr = tid + k * NT, q = q0 + lr;
            mark[k] = -1;
            if (lr >= ENCW_RB || q >= qe) continue;
            const int fe = encw_div(q, sp.rows_env, m_rows), ag = q - fe * sp.rows_env, t = encw_div(fe, a.E, m_E), r = q - t * a.R;
            const int32_t* st = a.snaps + (size_t)t * a.snap_words;
            const unsigned wa = ent[(fe - fe0) * sp.total + ag];
            const bool lv = sp.live(st, (size_t)r);
            if (lv && has_pos) {
                const int pos = sp.pos(wa), gb = (pos >> 5) - gb0;
                if ((unsigned)gb < (unsigned)MBP) {
                    mark[k] = ((((lr >> 4) * MBP + gb) * 64 + (pos & 31) + 32 * ((lr >> 3) & 1)) * 8 + (lr & 7));
                    afh[mark[k]] = 0x3F80;
                }
            }
            if (has_slots) {
                if ((unsigned)(a.nslots - s0) < 32u) Wt[(a.nslots - s0) * ENCW_LDW + lr] = 1.0f;       // the bias row: every row
                if (lv)
                    sp.self(st, (size_t)r, ag, wa, [&](int slot, float wgt) {
                        if ((unsigned)(slot - s0) < 32u) atomicAdd(&Wt[(slot - s0) * ENCW_LDW + lr], wgt);
                    });
            }
        }
        if (has_slots) {                                          // the other entities: row lr = it % RB against p = it / RB, + PG, ...
            const int PG = NT >= ENCW_RB ? NT / ENCW_RB : 1;
            for (int it = tid; it < ENCW_RB * PG; it += NT) {
                const int lr = it & (ENCW_RB - 1), pg = it / ENCW_RB, q = q0 + lr;
                if (q >= qe) continue;
                const int fe = encw_div(q, sp.rows_env, m_rows), ag = q - fe * sp.rows_env;
                if (!sp.live_always) {
                    const int t = encw_div(fe, a.E, m_E);
                    if (!sp.live(a.snaps + (size_t)t * a.snap_words, (size_t)(q - t * a.R))) continue;
                }
                const unsigned* ee = ent + (fe - fe0) * sp.total;
                const unsigned wa = ee[ag];
                for (int p = pg; p < sp.total; p += PG) {
                    const int slot = sp.pair(wa, ee[p], ag, p);
                    if (p != ag && (unsigned)(slot - s0) < 32u) atomicAdd(&Wt[(slot - s0) * ENCW_LDW + lr], 1.0f);
                }
            }
        }
        __syncthreads();
        const int nks = (min(qe - q0, ENCW_RB) + 15) / 16;
        auto kstep = [&](float (&gv_)[8], int ks) {
            ps_u32x4 bf[3];
            ps_split_frag(ps_f32x4{ gv_[0], gv_[1], gv_[2], gv_[3] }, ps_f32x4{ gv_[4], gv_[5], gv_[6], gv_[7] }, bf);
            fetch(gv_);                                           // the rows of the K step four ahead (past the range: zeros)
            if (has_pos) {
                ps_u32x4 af[MBP];
#pragma unroll
                for (int b = 0; b < MBP; ++b) af[b] = afr[(ks * MBP + b) * 64 + lane];
#pragma unroll
                for (int p = 2; p >= 0; --p)                      // (least significant term first; consecutive products independent)
#pragma unroll
                    for (int b = 0; b < MBP; ++b)
                        accp[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(encw_bf16x8, af[b]),
                                                                          __builtin_bit_cast(encw_bf16x8, bf[p]), accp[b], 0, 0, 0);
            }
            if (has_slots) {
                const float* wr = Wt + li * ENCW_LDW + ks * 16 + lh * 8;
                const ps_f32x4 w0 = *reinterpret_cast<const ps_f32x4*>(wr), w1 = *reinterpret_cast<const ps_f32x4*>(wr + 4);
                if (sp.slots_exact_bf16) {                        // small integer counts: one bf16 term holds them
                    const ps_u32x4 af = { ps_hi_pair(ps_f32x2{ w0[0], w0[1] }), ps_hi_pair(ps_f32x2{ w0[2], w0[3] }),
                                          ps_hi_pair(ps_f32x2{ w1[0], w1[1] }), ps_hi_pair(ps_f32x2{ w1[2], w1[3] }) };
#pragma unroll
                    for (int pb = 2; pb >= 0; --pb)
                        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(encw_bf16x8, af),
                                                                       __builtin_bit_cast(encw_bf16x8, bf[pb]), accs, 0, 0, 0);
                } else {
                    ps_u32x4 af[3];
                    ps_split_frag(w0, w1, af);
#pragma unroll
                    for (int pb = 2; pb >= 0; --pb)
#pragma unroll
                        for (int pa = 2; pa >= 0; --pa)
                            accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(encw_bf16x8, af[pa]),
                                                                           __builtin_bit_cast(encw_bf16x8, bf[pb]), accs, 0, 0, 0);
                }
            }
        };
#pragma unroll 1
        for (int ks = 0; ks < nks; ks += 4) {
            kstep(g0, ks);
            if (ks + 1 < nks) kstep(g1, ks + 1);
            if (ks + 2 < nks) kstep(g2, ks + 2);
            if (ks + 3 < nks) kstep(g3, ks + 3);
        }
    }
    // block b, register reg, lane (li, lh): output row m = 32 block + (reg & 3) + 8 (reg >> 2) + 4 lh, column `col`
    float* Pg = a.Ppart + (size_t)blockIdx.x * a.npos * a.H;
    float* Dg = a.Dpart + (size_t)blockIdx.x * (a.nslots + 1) * a.H;
#pragma unroll
    for (int b = 0; b < MBP; ++b)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = 32 * (gb0 + b) + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (m >= a.npos) continue;
            float* dst = Pg + (size_t)m * a.H + col;
            *dst = a.accumulate ? *dst + accp[b][reg] : accp[b][reg];
        }
    if (has_slots)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = s0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (m > a.nslots) continue;
            float* dst = Dg + (size_t)m * a.H + col;
            *dst = a.accumulate ? *dst + accs[reg] : accs[reg];
        }
}

// sum over partials [k0, k1) of P (the per-env form has one, atomically accumulated)
__device__ __forceinline__ float enc_bwd_psum(const float* __restrict__ P, int k0, int k1, size_t stride, size_t idx)
{
    float v = 0.f;
    for (int k = k0; k < k1; ++k) v += P[(size_t)k * stride + idx];
    return v;
}
// The expand stage gives the ENCB_PSPLIT = 4 quarter-waves of a wavefront a share of the partials of the same 16
// (pos, column) items each and folds them with two shuffles (one thread walking 128 partials 200 KB apart is a latency
// chain: 33 us per call at PP-hard; one atomic per share instead multiplies the atomics).  Thread i of the A part:
// item = 16 (i / 64) + (i % 16), share = (i % 64) / 16; nA = 64 ceil(items / 16) threads, whole wavefronts.
constexpr int ENCB_PSPLIT = 4;
__device__ __forceinline__ float enc_bwd_pfold(const float* __restrict__ P, int np, size_t stride, long long i, long long items,
                                               long long* item_out)
{
    const long long item = 16 * (i >> 6) + (i & 15);
    const int share = (int)((i & 63) >> 4), pper = (np + ENCB_PSPLIT - 1) / ENCB_PSPLIT;
    float v = 0.f;
    if (item < items) v = enc_bwd_psum(P, min(np, share * pper), min(np, (share + 1) * pper), stride, (size_t)item);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    *item_out = (share == 0 && item < items) ? item : -1;
    return v;
}
__host__ __device__ inline long long enc_bwd_pfold_threads(long long items) { return 64 * ((items + 15) / 16); }

}  // namespace ic3
