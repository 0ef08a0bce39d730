#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ac
mkdir -p $O
L=$PWD/ic3net_amd/csrc
IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$PWD/$O/trace_pp_hard.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_trace.json 2> $O/bench_trace.err
tail -c 600 $O/bench_trace.json
python tools/analyze_trace.py $O/trace_pp_hard.csv | tee $O/trace_analysis.txt
IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$PWD/$O/trace_pp_hard_noobs.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --no-dense-obs > $O/bench_trace_noobs.json 2> $O/bench_trace_noobs.err
python tools/analyze_trace.py $O/trace_pp_hard_noobs.csv > $O/trace_analysis_noobs.txt
