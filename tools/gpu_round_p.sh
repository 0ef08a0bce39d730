#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
for m in 0 4 8 12 16 1 2; do
  IC3_MB_OBS=1 IC3_PS_DEBUG=$m timeout 200 python tools/microbench_policy_step.py pp_hard 8192 40 mega >> $O/ablation_obs.txt 2>&1
done
for m in 0 4 8; do
  IC3_PS_DEBUG=$m timeout 200 python tools/microbench_policy_step.py pp_hard 8192 40 mega >> $O/ablation_noobs.txt 2>&1
done
echo "with obs (nt):"; grep median $O/ablation_obs.txt; echo "no obs:"; grep median $O/ablation_noobs.txt
