#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3x
mkdir -p $O
timeout 900 python -m pytest tests/test_trainer_gpu.py -m gpu -x -q -p no:cacheprovider -k "native_update or compute_grad or train_batch" > $O/pytest.log 2>&1
tail -n 25 $O/pytest.log
