// TEST INFRASTRUCTURE — the rest of libic3rollout_host.so: the C ABI of include/ic3_rollout.h on a CPU (`device = -1`).
//
// tests/host/Makefile compiles the PRODUCT's own sources — ic3net_amd/csrc/{ic3_api, pp_kernels, tj_kernels, policy_ops,
// episode_kernels}.hip and tj_tables.cpp, unmodified — as C++ against tests/host/shim/hip/hip_runtime.h, which runs every
// hipLaunchKernelGGL as workgroups of cooperatively scheduled lane fibers over host memory.  The result exports the same entry points
// as libic3rollout.so with the same argument conventions ("identical entry points exist in the CPU build (device = -1,
// hipStream_t ignored) so the same tests drive both"): tests/test_host_abi_cpu.py drives the
// reference's golden trajectories, the observation / encoder / encoder-backward kernels and the episode bookkeeping
// through it with numpy buffers, also under ASan / UBSan (tools/host_asan.sh).
//
// What is NOT in the host build: the kernels written against the matrix cores (policy_step.hip, gates_bwd.hip,
// commnet_fwd.hip — MFMA builtins and gfx950 inline assembly).  Their entry points are defined below and answer -ENOSYS,
// like the GPU library does for a shape it does not support.  Nothing under ic3net_amd/ loads this library, it never calls
// into oracle/, and the GPU library refuses device = -1: this is not a CPU path of the product.
#include "ic3_common.hpp"

using namespace ic3;

namespace {
int no_matrix_cores(const char* who)
{
    return fail(-38, std::string(who) + ": written against the gfx950 matrix cores, not part of the host build (tests/host)");
}
}  // namespace

extern "C" {

int ic3_policy_step_supported(const ic3_env*, int) { return 0; }
int ic3_lstm_gates_backward_supported(int) { return 0; }
int ic3_commnet_forward_supported(int, int) { return 0; }

int ic3_policy_pack(const float*, const float*, const float*, float*, float*, int, ic3_stream) { return no_matrix_cores("ic3_policy_pack"); }
int ic3_policy_pack_split(const float*, const float*, void*, int, ic3_stream) { return no_matrix_cores("ic3_policy_pack_split"); }
int ic3_policy_forward(const ic3_policy*, const float*, int, int, float*, float*, const int32_t*, const int32_t*, float*,
                       ic3_stream)
{
    return no_matrix_cores("ic3_policy_forward");
}
int ic3_policy_step(ic3_env*, const ic3_policy*, float*, float*, const int32_t*, const int32_t*, float*, int32_t*, float*,
                    float*, int32_t*, int32_t*, int32_t*, ic3_stream)
{
    return no_matrix_cores("ic3_policy_step");
}
int ic3_lstm_gates_backward(float*, int, const float*, const float*, const void*, const float*, const float*, const float*,
                            const float*, float*, float*, float*, int, int, int, ic3_stream)
{
    return no_matrix_cores("ic3_lstm_gates_backward");
}
int ic3_commnet_pack(const float*, const float*, float*, int, ic3_stream) { return no_matrix_cores("ic3_commnet_pack"); }
int ic3_commnet_forward(const float*, int, int, int, int, const float*, const float*, const float*, const float*,
                        const int32_t*, int, int, int, const int32_t*, const int32_t*, float*, float*, ic3_stream)
{
    return no_matrix_cores("ic3_commnet_forward");
}

}  // extern "C"
