#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3z
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_gpu.py -m gpu -x -q -p no:cacheprovider -k "equals_the_launch_chain" > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
