#!/bin/bash
# The product's device code on the host under AddressSanitizer + UBSan (no GPU needed):
#   tests/host/libic3host_asan.so          env_device.hpp's wave-level functions, the reference's golden PP / TJ trajectories
#                                          (tests/test_host_build_cpu.py, first two envs / episodes of every fixture)
#   tests/host/libic3rollout_host_asan.so  the product's .hip sources themselves behind the C ABI (device = -1): golden
#                                          trajectories through the reset / step / observation kernels, sparse encoder and
#                                          its backward, policy_ops.hip, episode_kernels.hip (tests/test_host_abi_cpu.py);
#                                          ic3_policy_step / _forward, ic3_commnet_forward, ic3_lstm_gates_backward at the
#                                          BASELINE shapes (tests/test_host_policy_step_cpu.py)
#   bash tools/host_asan.sh [log]
set -u
cd "$(dirname "$0")/.."
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$($CXX -print-file-name=libclang_rt.asan-x86_64.so)
UB=$($CXX -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
[ -f "$RT" ] || { echo "no ASan runtime at $RT"; exit 1; }
make -C tests/host asan || exit 1
LOG=${1:-/dev/stdout}
{
  echo "# $(date -u +%F) host ASan+UBSan run of the product's device code (tools/host_asan.sh)"
  echo "# runtime: $RT"
  LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    IC3_HOST_ASAN=1 python -m pytest tests/test_host_build_cpu.py tests/test_host_abi_cpu.py tests/test_host_policy_step_cpu.py -q -p no:cacheprovider 2>&1
  echo "# exit $?"
} > "$LOG"
tail -5 "$LOG"
