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
