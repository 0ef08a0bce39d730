#!/bin/bash
# round 3, call ZB: rocprofv3 kernel stats of the gate_split experiment (agreement with the in-bench dispatch events)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3zb
rm -rf $O; mkdir -p $O
cd $R
S="--steps 40 --warmup 8 --no-cpu-baseline --gate-split 1"
for w in tj_hard tj_medium pp_hard; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python bench.py $S --workload $w > $O/kt_$w.log 2>&1
  f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1)
  cp $f $O/bench_${w}_EXPERIMENT_gate_split_kernel_stats.csv
  head -2 $O/bench_${w}_EXPERIMENT_gate_split_kernel_stats.csv | cut -c1-160
  grep -o '"avg_launch_ms": [0-9.]*' $O/kt_$w.log | head -1
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; rm -rf $O/kt_*/
