#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_rccl_world1_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
timeout 600 python tools/exp/microbench_bptt_gemms.py > $O/gemms.txt 2>&1; grep -v amdgpu $O/gemms.txt
timeout 900 python tools/bench_train.py 8192 3 native > $O/train_8192_native.txt 2>&1; tail -n 1 $O/train_8192_native.txt
