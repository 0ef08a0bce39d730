#!/bin/bash
export TMPDIR=/tmp
python tools/exp/microbench_gates_bwd.py 2>&1 | tail -1
IC3_ROLLOUT_LIB=$PWD/ic3net_amd/csrc/libic3rollout_gbagpr.so python tools/exp/microbench_gates_bwd.py 2>&1 | tail -1
IC3_ROLLOUT_LIB=$PWD/ic3net_amd/csrc/libic3rollout_gbagpr.so timeout 300 python -m pytest tests/test_gates_backward_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
python tools/exp/microbench_gates_bwd.py 163840 128 2>&1 | tail -1
python tools/exp/microbench_gates_bwd.py 81920 64 2>&1 | tail -1
python tools/exp/microbench_gates_bwd.py 40960 256 2>&1 | tail -1
