#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3zc
mkdir -p $O
R=$PWD
for W in tj_hard pp_hard; do
  IC3_ROLLOUT_LIB=$R/ic3net_amd/csrc/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace_$W.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 --workload $W --gate-split 1 > /dev/null 2>&1
  python tools/analyze_trace.py $O/trace_$W.csv > $O/phase_trace_${W}_gate_split.txt 2>&1; rm -f $O/trace_$W.csv
  head -24 $O/phase_trace_${W}_gate_split.txt
done
