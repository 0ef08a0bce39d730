#!/bin/bash
# The product's device functions (ic3net_amd/csrc/env_device.hpp) on the host under AddressSanitizer + UBSan:
# builds tests/host/libic3host_asan.so and drives the reference's golden PP / TJ trajectories through it
# (tests/test_host_build_cpu.py, first two envs / episodes of every fixture).  No GPU needed.
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
  echo "# $(date -u +%F) host ASan+UBSan run of env_device.hpp (tools/host_asan.sh)"
  echo "# runtime: $RT"
  LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    IC3_HOST_ASAN=1 python -m pytest tests/test_host_build_cpu.py -q -p no:cacheprovider 2>&1
  echo "# exit $?"
} > "$LOG"
tail -5 "$LOG"
