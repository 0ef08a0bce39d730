#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
./tools/exp/ws_probe > $O/ws_probe.txt 2>&1
timeout 600 python -m pytest tests/test_policy_step_gpu.py -q --maxfail=30 -p no:cacheprovider -k "trainer_takes" > $O/tests.log 2>&1
for k in 0 4 8 12 16; do
  IC3_PS_SKEW=$k timeout 200 python tools/microbench_policy_step.py pp_hard 8192 40 mega >> $O/skew.txt 2>&1
done
cat $O/ws_probe.txt; grep -E "AssertionError|passed|failed" $O/tests.log | head; grep "median" $O/skew.txt
