#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3ze
mkdir -p $O
timeout 600 python -m pytest tests/test_gates_backward_gpu.py tests/test_trainer_gpu.py -m gpu -x -q -p no:cacheprovider -k "gates_backward or native" 2>&1 | tail -3
python tools/exp/microbench_gates_bwd.py 2>&1 | tail -1 | tee $O/gates_bwd.txt
SPLIT=1 python tools/exp/microbench_gates_bwd.py 2>&1 | tail -1 | tee -a $O/gates_bwd.txt
python tools/exp/microbench_gates_bwd.py 163840 2>&1 | tail -1 | tee -a $O/gates_bwd.txt
SPLIT=1 python tools/exp/microbench_gates_bwd.py 163840 2>&1 | tail -1 | tee -a $O/gates_bwd.txt
timeout 600 python tools/bench_train.py 8192 4 native tj_hard 1 0 2>&1 | tail -1 | tee $O/train.txt
timeout 600 python tools/bench_train.py 8192 4 native tj_hard 1 1 2>&1 | tail -1 | tee -a $O/train.txt
timeout 600 python tools/bench_train.py 8192 4 native pp_hard 1 0 2>&1 | tail -1 | tee -a $O/train.txt
timeout 600 python tools/bench_train.py 8192 4 native pp_hard 1 1 2>&1 | tail -1 | tee -a $O/train.txt
