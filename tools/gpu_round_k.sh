#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02k
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
run fused_nohalf         IC3_PS_HALF=0 $B
run noobs                $B --no-dense-obs
run noobs_nohalf         IC3_PS_HALF=0 $B --no-dense-obs
run tj_hard              $B --workload tj_hard
run tj_hard_nohalf       IC3_PS_HALF=0 $B --workload tj_hard
run tj_medium            $B --workload tj_medium
run pp_easy              $B --workload pp_easy
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/tests_ps.log 2>&1
grep -E "passed|failed|FAILED" $O/tests_ps.log | tail -n 6
