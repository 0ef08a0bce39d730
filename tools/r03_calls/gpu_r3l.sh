#!/bin/bash
# round 3: row-parallel encoder backward + fused gate recompute / cell backward — tests, then the update at E = 8192
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
timeout 900 python -m pytest tests/test_gates_backward_gpu.py tests/test_encode_backward_gpu.py tests/test_trainer_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 15 $O/pytest.log
timeout 600 python tools/bench_train.py 8192 4 native > $O/train_8192.txt 2>&1; tail -n 2 $O/train_8192.txt
timeout 600 python tools/profile_train_native.py 8192 > $O/train_8192_profile.txt 2>&1; grep -E "ic3::|Cijk|Memcpy|Memset|reduce_kernel|Self CUDA time" $O/train_8192_profile.txt | cut -c1-60,150-230 | head -40
