#!/bin/bash
# cache policy of ic3_lstm_gates_backward_given's streams (IC3_GB_NT variants) inside a PP-hard update
for v in "" gbnt1 gbnt2 gbnt3 "" gbnt3; do
  lib=ic3net_amd/csrc/libic3rollout${v:+_$v}.so
  echo "== $lib"
  IC3_ROLLOUT_LIB=$PWD/$lib python tools/bench_train.py 8192 4 native pp_hard 2>&1 | grep train_batch
done
