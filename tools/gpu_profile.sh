#!/bin/bash
# Profiles of the default bench command (round 2): kernel trace + stats, then the two HBM PMC passes, into gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_multirank_gpu.py -q -p no:cacheprovider > $O/tests_multirank.log 2>&1
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
timeout 600 python bench.py --steps 160 --warmup 16 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B > $O/pmc_f.log 2>&1
for w in tj_hard tj_medium pp_easy; do timeout 300 $B --workload $w > $O/bench_$w.json 2> /dev/null; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sep -- $B --fused-obs 0 > $O/kt_sep.log 2>&1
timeout 300 $B --fused-obs 0 > $O/bench_separate.json 2>/dev/null
timeout 300 $B --mega 0 --time-kernels 0 > $O/bench_chain.json 2>/dev/null
IC3_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $O/roctx -- python bench.py --steps 8 --warmup 2 --nenvs 1024 --no-cpu-baseline > $O/roctx.log 2>&1
find $O -name "*.csv" | head -40; du -sh $O; tail -3 $O/tests_multirank.log; cat $O/bench_default.json
