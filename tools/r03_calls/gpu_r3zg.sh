#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3zg
mkdir -p $O
timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt | tail -12
