#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}
    print("%-28s %.4f ms/step  %.1f M/s | hbm-kernel %.4f ms %.0f GB/s | %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0), d['config']['launch']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
run fused                $B
run noobs                $B --no-dense-obs
run tj_hard              $B --workload tj_hard
run tj_medium            $B --workload tj_medium
run separate             $B --fused-obs 0
run chain_graph          $B --mega 0 --time-kernels 0
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/tests_ps.log 2>&1
grep -E "passed|failed|FAILED" $O/tests_ps.log | tail -n 6
