#!/bin/bash
# full GPU suite + smoke + update-half evidence on the current code
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
timeout 600 python tools/bench_train.py 8192 4 native > $O/train_batch_8192_native.txt 2>&1; tail -n 1 $O/train_batch_8192_native.txt
timeout 600 python tools/bench_train.py 1024 4 native > $O/train_batch_1024_native.txt 2>&1; tail -n 1 $O/train_batch_1024_native.txt
timeout 600 python tools/profile_train_native.py 8192 > $O/train_batch_8192_native_profile.txt 2>&1; grep -E "Self CUDA time" $O/train_batch_8192_native_profile.txt
python tools/exp/microbench_encode_bwd.py 8192 pp_hard 2>&1 | tail -1 | tee $O/encode_bwd.txt
python tools/exp/microbench_encode_bwd.py 8192 tj_hard 2>&1 | tail -1 | tee -a $O/encode_bwd.txt
python tools/exp/microbench_encode_bwd.py 8192 tj_medium 2>&1 | tail -1 | tee -a $O/encode_bwd.txt
python tools/exp/microbench_gates_bwd.py 2>&1 | tail -1 | tee $O/gates_bwd.txt
python tools/exp/microbench_gates_bwd.py 163840 2>&1 | tail -1 | tee -a $O/gates_bwd.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
