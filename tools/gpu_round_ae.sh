#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ag
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print("%-28s %.4f ms/step  %.1f M/s | hbm-kernel %.4f ms %.0f GB/s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0)))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16"
L=$PWD/ic3net_amd/csrc
timeout 1200 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_gpu.py tests/test_trainer_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -n 2 $O/tests.log
run warm    $B
run base_1  $B
run base_2  $B
run tj_hard $B --workload tj_hard
run tj_medium $B --workload tj_medium
run s20_w5  python bench.py --no-cpu-baseline --steps 20 --warmup 5
IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$PWD/$O/trace_pp_hard.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 > $O/bench_trace.json 2> $O/bench_trace.err
python tools/analyze_trace.py $O/trace_pp_hard.csv > $O/trace_analysis.txt; head -24 $O/trace_analysis.txt
