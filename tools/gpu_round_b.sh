#!/bin/bash
# GPU call B: re-run the failing tests, determinism, and the phase ablations of policy_step_kernel
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_main_gpu.py tests/test_policy_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt
for m in 0 1 2 4 8 16 3 7 15 31; do
  IC3_PS_DEBUG=$m timeout 200 python tools/microbench_policy_step.py pp_hard 8192 40 >> $O/ablation.txt 2>&1
done
timeout 300 python bench.py --steps 160 --warmup 16 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|FAILED" $O/tests.log | tail -n 20; cat $O/summary.txt; grep IC3_PS $O/ablation.txt; cat $O/bench.json
