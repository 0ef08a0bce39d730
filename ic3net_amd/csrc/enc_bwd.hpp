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
#ifndef IC3_ENCB_NOATOMIC   // (timing experiment, round 3: PP-hard E = 8192 takes 212 us per call with these atomics, 81 us without)
        if (p >= 0) {
            float* dst = P + (size_t)p * H + 4 * c4;
            if (v.x != 0.f) atomicAdd(dst + 0, v.x);
            if (v.y != 0.f) atomicAdd(dst + 1, v.y);
            if (v.z != 0.f) atomicAdd(dst + 2, v.z);
            if (v.w != 0.f) atomicAdd(dst + 3, v.w);
        }
#endif
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

}  // namespace ic3
