#!/bin/bash
# round 3, call V: does the default (native) update path still learn?  PP-easy, 300 updates, both update paths
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
timeout 900 python tools/train_sanity.py 300 native > $O/train_sanity_native.txt 2>&1; tail -n 6 $O/train_sanity_native.txt
timeout 900 python tools/train_sanity.py 300 autograd > $O/train_sanity_autograd.txt 2>&1; tail -n 4 $O/train_sanity_autograd.txt
