#!/bin/bash
# round 3, call E: native update half (tests + throughput), new pacing defaults, dropped-store ablation
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_policy_step_gpu.py tests/test_main_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for W in pp_hard tj_hard tj_medium; do
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  run ${W}_default $B
  run ${W}_drop IC3_ROLLOUT_LIB=$L/libic3rollout_drop.so $B
  run ${W}_noobs $B --no-dense-obs
done
for mode in native autograd; do
  timeout 600 python tools/bench_train.py 1024 3 $mode > $O/train_1024_$mode.txt 2>&1; tail -n 1 $O/train_1024_$mode.txt
done
timeout 900 python tools/bench_train.py 8192 3 native > $O/train_8192_native.txt 2>&1; tail -n 1 $O/train_8192_native.txt
TUNE=0 timeout 900 python tools/bench_train.py 8192 3 native > $O/train_8192_native_notune.txt 2>&1; tail -n 1 $O/train_8192_native_notune.txt
