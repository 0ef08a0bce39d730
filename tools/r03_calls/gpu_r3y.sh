#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
timeout 300 ./tools/exp/bf16x9_probe > $O/bf16x9_probe.txt 2>&1; cat $O/bf16x9_probe.txt
