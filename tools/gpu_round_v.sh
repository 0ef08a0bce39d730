#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
timeout 600 python tools/exp/two_stream.py pp_hard 8192 2>&1 | grep -v amdgpu.ids | tee $O/two_stream.txt
for E in 384 768 1536; do for OB in 0 1; do IC3_MB_OBS=$OB timeout 200 python tools/microbench_policy_step.py pp_hard $E 48 2>/dev/null | grep "ic3_policy_step" | sed "s/^/obs=$OB /"; done; done | tee $O/small_e.txt
