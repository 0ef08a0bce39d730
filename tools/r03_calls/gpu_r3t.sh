#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
timeout 600 python tools/bench_train.py 8192 3 native tj_hard > $O/train_batch_8192_native_tj_hard.txt 2>&1; tail -n 1 $O/train_batch_8192_native_tj_hard.txt
timeout 600 python tools/bench_train.py 8192 3 native tj_medium > $O/train_batch_8192_native_tj_medium.txt 2>&1; tail -n 1 $O/train_batch_8192_native_tj_medium.txt
timeout 600 python tools/bench_train.py 8192 4 native pp_hard 0 > $O/train_batch_8192_native_no_dense_obs.txt 2>&1; tail -n 1 $O/train_batch_8192_native_no_dense_obs.txt
timeout 600 python tools/bench_train.py 8192 4 native pp_hard 1 > $O/train_batch_8192_native.txt 2>&1; tail -n 1 $O/train_batch_8192_native.txt
