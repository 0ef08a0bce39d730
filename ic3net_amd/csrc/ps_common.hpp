// ps_common.hpp — what policy_step.hip (the one-launch rollout step) and policy_pack.hip (its weight packing and the
// gate-product probe) share: vector types, the compiler-visible buffer loads, the fp32 matrix instruction, and the EXACT split
// of an fp32 value into three bf16 terms (ic3_policy.gate_split).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ic3 {

typedef float ps_f32x4 __attribute__((ext_vector_type(4)));
typedef float ps_f32x16 __attribute__((ext_vector_type(16)));
typedef int ps_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_acc(ps_f32x16& acc, float x, float y)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
}

// ---- vector-memory bookkeeping --------------------------------------------------------------------------------------
// A wave has ONE counter (vmcnt) for its outstanding loads AND stores, they complete in issue order, and s_waitcnt
// takes an immediate.  Round 2 issued the obs zero stores as inline asm behind a run-time count: invisible to the
// compiler, whose `s_waitcnt vmcnt(n)` in front of each MFMA group therefore counted only the weight loads — with
// stores in between, "at most n operations outstanding" turned into "the zero stores issued a moment ago have been
// acknowledged" (the gate loop ran 12 % over its MFMA time, every load behind the loop first drained the store queue).
// Now every vector-memory operation of the kernel is a compiler-visible builtin and the zero stores are issued
// UNCONDITIONALLY — the hardware range check of their buffer descriptor drops the ones past the tile's slice
// (tools/exp/buf_probe.hip: VGPR and SGPR offsets both take part in the check) — so the number of operations between
// any load and its first use is a compile-time property of the program and the compiler's waits are exact.
// (A first version kept inline-asm loads with hand-written waits tied to their registers by "+v" operands: the compiler
// is free to COPY such a register in front of the wait, and did — stale weights whenever the L2 was cold.)
typedef unsigned int ps_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ ps_f32x4 buf_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- ic3_policy.gate_split (the default since round 4; DESIGN.md section 0, tools/exp/bf16x9_probe.hip): the gate
// product with every fp32 operand split EXACTLY into three bf16 terms (x = x1 + x2 + x3, round-to-nearest-even splits,
// exact residuals) and all nine cross products on v_mfma_f32_32x32x16_bf16 — each product exact in fp32, fp32
// accumulation.  Weights: pre-split planes in fragment order (ic3_policy_pack_split); activations: the fp32 A tile stays
// in LDS as it is and every wave splits the 8 values of its row per 16 k-steps in registers.
typedef __bf16 ps_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned ps_bf16_rne(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void ps_split3(float x, unsigned& x1, unsigned& x2, unsigned& x3)
{
    x1 = ps_bf16_rne(x);
    const float r1 = x - __builtin_bit_cast(float, x1 << 16);
    x2 = ps_bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, x2 << 16);
    x3 = ps_bf16_rne(r2);
}
// 8 consecutive fp32 of one row -> the three bf16 A fragments of a 32x32x16 MFMA.  Pairwise through v_cvt_pk_bf16_f32
// (round-to-nearest-even in hardware, the pair comes out packed): 9 vector instructions per pair.
typedef float ps_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ps_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ps_split_pair(ps_f32x2 x, unsigned& p1, unsigned& p2, unsigned& p3)
{
    const ps_bf16x2 h1 = __builtin_convertvector(x, ps_bf16x2);
    const ps_f32x2 r1 = x - __builtin_convertvector(h1, ps_f32x2);
    const ps_bf16x2 h2 = __builtin_convertvector(r1, ps_bf16x2);
    const ps_f32x2 r2 = r1 - __builtin_convertvector(h2, ps_f32x2);
    const ps_bf16x2 h3 = __builtin_convertvector(r2, ps_bf16x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, h2);
    p3 = __builtin_bit_cast(unsigned, h3);
}
// The same split in stages, for the gate loop's software pipeline: the most significant plane of a pair, then
// `r -= float(h)` and the next plane of what is left (ps_split_pair = ps_hi_pair, ps_next_pair, ps_next_pair).
__device__ __forceinline__ unsigned ps_hi_pair(ps_f32x2 x)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, ps_bf16x2));
}
__device__ __forceinline__ unsigned ps_next_pair(ps_f32x2& r, unsigned h)
{
    r = r - __builtin_convertvector(__builtin_bit_cast(ps_bf16x2, h), ps_f32x2);
    return __builtin_bit_cast(unsigned, __builtin_convertvector(r, ps_bf16x2));
}
__device__ __forceinline__ void ps_split_frag(ps_f32x4 x0, ps_f32x4 x1, ps_u32x4 (&out)[3])
{
    unsigned p[3][4];
    ps_split_pair(ps_f32x2{ x0[0], x0[1] }, p[0][0], p[1][0], p[2][0]);
    ps_split_pair(ps_f32x2{ x0[2], x0[3] }, p[0][1], p[1][1], p[2][1]);
    ps_split_pair(ps_f32x2{ x1[0], x1[1] }, p[0][2], p[1][2], p[2][2]);
    ps_split_pair(ps_f32x2{ x1[2], x1[3] }, p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = ps_u32x4{ p[pl][0], p[pl][1], p[pl][2], p[pl][3] };
}

}  // namespace ic3
