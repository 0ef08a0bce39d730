// obs_fill.hip — the zero background of the dense observation rows as a launch of its own (gfx950).
//
// The rows ic3_policy_step hands back (the `state` of trainer.py:49; predator_prey_env.py:188-210 /
// traffic_junction_env.py:321-366 flattened by env_wrappers.py:88-100) are ~98 % zeros.  Written from inside the policy
// launch, every 1 KiB of zeros is an instruction in the stream of a wave that also issues the matrix work, and it shares
// that wave's ONE in-order memory counter with the weight loads.  ic3_obs_prefill moves the zeros
// into waves that do nothing else: a workgroup of this kernel needs 8 vector registers per lane and no LDS, so it fits into
// what two resident policy_step workgroups leave of a CU (2 x 248 of 512 registers per SIMD lane) and runs BESIDE them
// on a second stream — filling the buffer of step t + 1 while step t computes.  The policy launch of step t + 1 then
// only patches the non-zero entries in (ic3_policy_step recognises the buffer).  Same bytes to HBM per step as before:
// every row is rewritten every step, by two launches instead of one.
//
// Geometry: one 16-byte non-temporal store per lane and trip, a wave store = 1 KiB contiguous, a workgroup (256 lanes)
// = 4 KiB per trip; a RESIDENT set of workgroups (one per CU: one wave per SIMD) walks the buffer.  The address arithmetic
// is all scalar (a buffer descriptor per trip whose range check clips the last block); the loop holds NO vector ALU
// instruction, so it does not compete with an MFMA stream for the SIMD's vector issue — measured, tools/exp/ws_probe.hip
// round 4: fp32 MFMA stream alone 0.486 ms, this store loop alone 0.225 ms (1.19 GB), both on the same CUs 0.477 ms;
// helper waves WITH vector address arithmetic (round 2's probe) were starved by the MFMA stream: 0.690 ms.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "ic3_common.hpp"

namespace ic3 {

typedef unsigned int of_u32x4 __attribute__((ext_vector_type(4)));

template <int NT_STORES>
__global__ __launch_bounds__(64) void obs_fill_kernel(float* __restrict__ base, unsigned int nblocks, unsigned int last_bytes,
                                                      int U, unsigned int start_stride, unsigned int trip_stride, int nap, int depth)
{
    // One wave per workgroup.  The 16-byte aligned body is `nblocks` blocks of 1 KiB (the last one holds last_bytes).  Trip
    // u of wave b covers block b * start_stride + u * trip_stride: (U, 1) = a contiguous slice per wave, (1, gridDim.x) =
    // all waves side by side in a window that moves through the buffer.  Everything is wave-uniform and 32-bit: the address
    // arithmetic and the loop control run on the scalar ALU (a 64-bit compare would be a vector instruction).
    // `nap`: quanta of 64 cycles the wave sleeps between two stores — the PACING.  A wave that stores as fast as the memory
    // pipeline takes them keeps the CU's vector-memory queue full, and the weight loads of the policy workgroups on the same
    // CU then wait behind ~a thousand cycles of stores per k-step (measured: the policy launch 4 x slower while the fill
    // runs); paced to the rate that finishes the buffer within the policy launch, the queue stays short.
    unsigned int bi = blockIdx.x * start_stride;
    of_u32x4 z = { 0u, 0u, 0u, 0u };
    IC3_OPAQUE_VGPR(z);                                                  // four registers, set once
    const int voff = (int)threadIdx.x * 16;
#pragma unroll 1
    for (int u = 0; u < U; ++u) {
        if (bi >= nblocks) break;
        const unsigned int mine = (bi == nblocks - 1u) ? last_bytes : 1024u;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(base) + ((unsigned long long)bi << 10), 0, mine, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(z, r, voff, 0, NT_STORES ? 2 : 0);   // past `mine`: dropped by the range check
        bi += trip_stride;
        // CLOSED-LOOP pacing: at most `depth` stores of this wave in flight.  Left to itself a wave keeps 64 of them
        // outstanding (the counter's range) — 64 MB across the chip, ~20 us of queue in front of every L2 channel, and the
        // policy launch's loads wait in those queues (measured: 4 x slower).
        switch (depth) {
        case 0: IC3_WAIT_VMEM_N(0); break;
        case 1: IC3_WAIT_VMEM_N(1); break;
        case 2: IC3_WAIT_VMEM_N(2); break;
        case 3: IC3_WAIT_VMEM_N(3); break;
        case 4: IC3_WAIT_VMEM_N(4); break;
        case 6: IC3_WAIT_VMEM_N(6); break;
        case 8: IC3_WAIT_VMEM_N(8); break;
        case 12: IC3_WAIT_VMEM_N(12); break;
        case 16: IC3_WAIT_VMEM_N(16); break;
        case 32: IC3_WAIT_VMEM_N(32); break;
        default: break;
        }
#pragma unroll 1
        for (int q = 0; q < nap; ++q) __builtin_amdgcn_s_sleep(1);
    }
}

__global__ void obs_fill_tail_kernel(float* __restrict__ p, int n)
{
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.0f;
}

}  // namespace ic3

using namespace ic3;

extern "C" int ic3_obs_set_prefilled(ic3_env* env, const float* obs)
{
    if (!env) return fail(-22, "ic3_obs_set_prefilled: null handle");
    env->prefilled_obs = obs;
    return 0;
}

extern "C" int ic3_obs_prefill(ic3_env* env, float* obs, ic3_stream stream)
{
    ic3::Range range_("ic3_obs_prefill");
    if (!env || !obs) return fail(-22, "ic3_obs_prefill: null argument");
    if (reinterpret_cast<uintptr_t>(obs) & 15) return fail(-22, "ic3_obs_prefill: obs must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long floats = (unsigned long long)env->dims.E * env->dims.N * env->dims.obs_dim;
    const unsigned long long bytes16 = (floats >> 2) << 4;
    // A RESIDENT set of waves (default: four per CU, one per SIMD) that walk the buffer, not one workgroup per 4 KiB:
    // measured in round 4 (profiles/r04/prefill_dispatch.txt) — 290 000 tiny workgroups next to the policy launch take every
    // wave slot that frees up, the dispatcher never collects the 2 slots per SIMD + 73 KB of LDS a policy workgroup needs,
    // and the two launches run one after the other instead of side by side.
    // Speed-only knobs: IC3_FILL_WAVES waves per CU x 100; IC3_FILL_MODE 0 contiguous slices, 1 moving window;
    // IC3_FILL_NAP sleep quanta (64 cycles) between two stores of a wave (-1: from env->fill_nap, see ic3_obs_set_fill_pace)
    static const int waves_env = getenv("IC3_FILL_WAVES") ? atoi(getenv("IC3_FILL_WAVES")) : 400;
    static const int mode = getenv("IC3_FILL_MODE") ? atoi(getenv("IC3_FILL_MODE")) : 0;
    static const int nap_env = getenv("IC3_FILL_NAP") ? atoi(getenv("IC3_FILL_NAP")) : -1;
    static const int depth = getenv("IC3_FILL_DEPTH") ? atoi(getenv("IC3_FILL_DEPTH")) : 4;
    static const int plain = getenv("IC3_FILL_PLAIN") ? atoi(getenv("IC3_FILL_PLAIN")) : 0;
    env->touch_obs(obs);
    if (bytes16) {
        static std::atomic<int> cus_of[64];                          // per device (a process may drive several GPUs); zero-initialised,
        int dev = 0, cus = 256;                                      // written once per device with the same value by whoever gets there
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            int known = cus_of[dev].load(std::memory_order_relaxed);
            if (!known) {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
                    known = prop.multiProcessorCount;
                    cus_of[dev].store(known, std::memory_order_relaxed);
                }
            }
            if (known) cus = known;
        }
        const unsigned long long blocks = (bytes16 + 1023ull) / 1024ull;
        if (blocks > 0x7fffffffull) return fail(-22, "ic3_obs_prefill: buffer too large for one launch");
        unsigned long long grid = (unsigned long long)std::max(1, cus * std::max(waves_env, 1) / 100);
        if (grid > blocks) grid = blocks;
        const unsigned int U = (unsigned int)((blocks + grid - 1) / grid);
        const unsigned int last_bytes = (unsigned int)(bytes16 - (blocks - 1) * 1024ull);
        const unsigned int start_stride = mode ? 1u : U;
        const unsigned int trip_stride = mode ? (unsigned int)grid : 1u;
        const int nap = nap_env >= 0 ? nap_env : env->fill_nap;
        if (plain)
            hipLaunchKernelGGL((obs_fill_kernel<0>), dim3((unsigned)grid), dim3(64), 0, s, obs, (unsigned int)blocks, last_bytes,
                               (int)U, start_stride, trip_stride, nap, depth);
        else
            hipLaunchKernelGGL((obs_fill_kernel<1>), dim3((unsigned)grid), dim3(64), 0, s, obs, (unsigned int)blocks, last_bytes,
                               (int)U, start_stride, trip_stride, nap, depth);
    }
    if (floats & 3)
        hipLaunchKernelGGL(obs_fill_tail_kernel, dim3(1), dim3(64), 0, s, obs + (floats & ~3ull), (int)(floats & 3));
    IC3_HIP(hipGetLastError());
    env->prefilled_obs = obs;
    return 0;
}
