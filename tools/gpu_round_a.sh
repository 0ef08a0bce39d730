#!/bin/bash
# GPU call A: correctness of the one-launch path first, then the whole GPU suite, then a first bench + microbench.
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/test_policy_step.log 2>&1
echo "policy_step rc=$?" > $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --deselect tests/test_policy_step_gpu.py > $O/test_all.log 2>&1
echo "suite rc=$?" >> $O/summary.txt
timeout 600 python bench.py --steps 160 --warmup 16 > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" >> $O/summary.txt
timeout 300 python bench.py --steps 160 --warmup 16 --time-kernels 0 --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
timeout 300 python bench.py --steps 160 --warmup 16 --mega 0 --no-cpu-baseline > $O/bench_chain.json 2> $O/bench_chain.err
timeout 300 python tools/microbench_policy_step.py pp_hard 8192 60 > $O/micro_pp_hard.txt 2>&1
timeout 300 python tools/microbench_policy_step.py tj_hard 8192 60 > $O/micro_tj_hard.txt 2>&1
timeout 300 python tools/microbench_policy_step.py tj_medium 8192 60 > $O/micro_tj_medium.txt 2>&1
tail -3 $O/test_policy_step.log $O/test_all.log; cat $O/summary.txt $O/bench_default.json $O/bench_graph.json $O/bench_chain.json $O/micro_*.txt
