#!/bin/bash
# The CPU oracle (oracle/ic3_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer over every golden vector
# recorded from the reference and the CPU property tests.  No GPU needed.
set -e
cd "$(dirname "$0")/.."
make -C oracle asan
export IC3_ORACLE_SO=$PWD/oracle/libic3oracle_asan.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python -m pytest tests/test_oracle_golden.py tests/test_properties_cpu.py -q -p no:cacheprovider "$@"
