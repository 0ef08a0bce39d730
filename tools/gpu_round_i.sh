#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02i
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
run auto               $B
run zb0_zl2            IC3_PS_ZB=0 IC3_PS_ZL=2 $B
run zb10_zl1           IC3_PS_ZB=10 IC3_PS_ZL=1 $B
run zb20_zl1           IC3_PS_ZB=20 IC3_PS_ZL=1 $B
run zb30_zl1           IC3_PS_ZB=30 IC3_PS_ZL=1 $B
run zb40_zl0           IC3_PS_ZB=40 IC3_PS_ZL=0 $B
run zb8_zl2            IC3_PS_ZB=8 IC3_PS_ZL=2 $B
run tj_hard_auto       $B --workload tj_hard
run tj_medium_auto     $B --workload tj_medium
timeout 600 python -m pytest tests/test_policy_step_gpu.py -q --maxfail=30 -p no:cacheprovider -k "equals_the_launch_chain or masks or determin" > $O/tests_ps.log 2>&1
grep -E "passed|failed" $O/tests_ps.log | tail -n 3
