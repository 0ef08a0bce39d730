// TEST INFRASTRUCTURE — libic3rollout_host.so: the C ABI of include/ic3_rollout.h on a CPU (`device = -1`).
//
// tests/host/Makefile compiles the PRODUCT's own sources — every ic3net_amd/csrc/*.hip and tj_tables.cpp, unmodified — as
// C++ against tests/host/shim/hip/hip_runtime.h, which runs every hipLaunchKernelGGL as workgroups of cooperatively
// scheduled lane fibers over host memory and executes the matrix-core builtins as cross-lane operations with the
// hardware's layouts.  The result exports the same entry points as libic3rollout.so with the same argument conventions
// ("identical entry points exist in the CPU build (device = -1, hipStream_t ignored) so the same tests drive both",
// SURVEY.md §8(b2)): tests/test_host_abi_cpu.py and tests/test_host_policy_step_cpu.py drive the reference's golden
// trajectories, the observation / encoder / encoder-backward kernels, the episode bookkeeping and the one-launch rollout
// step through it with numpy buffers, also under ASan / UBSan (tools/host_asan.sh).
//
// Nothing under ic3net_amd/ loads this library, it never calls into oracle/, and the GPU library refuses device = -1:
// this is not a CPU path of the product.  This translation unit only anchors the library (every symbol comes from the
// product's sources).
#include "ic3_common.hpp"

// ---- self-tests of the stand-in runtime's guards (tests/test_host_abi_cpu.py runs them in subprocesses: both must abort) ----
namespace {
__global__ void selftest_lds_overrun_kernel(int words)
{
    IC3_DYNAMIC_LDS(int32_t, smem);
    if (threadIdx.x == 0) smem[words] = 1;   // the first word BEHIND what the launch asked for
}
__global__ void selftest_barrier_kernel()
{
    // every lane waits for a partner value that only arrives through a shuffle the odd lanes never reach
    if ((threadIdx.x & 1) == 0) (void)__shfl((int)threadIdx.x, 1);
    else
        for (;;) __syncthreads();
}
}  // namespace

extern "C" int ic3_host_selftest_lds_overrun(void)
{
    hipLaunchKernelGGL(selftest_lds_overrun_kernel, dim3(1), dim3(64), 64 * sizeof(int32_t), nullptr, 64);
    return 0;   // not reached: the runtime aborts behind the workgroup ("wrote past the dynamic LDS")
}
extern "C" int ic3_host_selftest_stuck_barrier(void)
{
    hipLaunchKernelGGL(selftest_barrier_kernel, dim3(1), dim3(64), 0, nullptr);
    return 0;   // not reached: "no lane can make progress"
}
