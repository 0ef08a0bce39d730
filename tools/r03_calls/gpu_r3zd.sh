#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3zd
mkdir -p $O
timeout 900 python tools/exp/gate_split_accuracy.py 2>&1 | grep -v amdgpu.ids | tee $O/gate_split_accuracy.txt
