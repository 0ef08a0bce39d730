#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_encode_backward_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
python tools/exp/microbench_encode_bwd.py 2>&1 | tail -1
for v in 1 2 7; do IC3_ROLLOUT_LIB=$PWD/ic3net_amd/csrc/libic3rollout_encb$v.so python tools/exp/microbench_encode_bwd.py 2>&1 | tail -1; done
python tools/exp/microbench_encode_bwd.py 8192 tj_hard 2>&1 | tail -1
python tools/exp/microbench_encode_bwd.py 8192 tj_medium 2>&1 | tail -1
python tools/exp/microbench_encode_bwd.py 1024 pp_hard 2>&1 | tail -1
