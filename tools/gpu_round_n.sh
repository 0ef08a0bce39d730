#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02n
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
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
run plain        $B
run sc1          IC3_PS_ZMODE=1 $B
run nt           IC3_PS_ZMODE=2 $B
run sc0sc1       IC3_PS_ZMODE=3 $B
run sc1_zb20     IC3_PS_ZMODE=1 IC3_PS_ZB=20 $B
run nt_zb20      IC3_PS_ZMODE=2 IC3_PS_ZB=20 $B
run tj_hard_sc1  IC3_PS_ZMODE=1 $B --workload tj_hard
run tj_hard_nt   IC3_PS_ZMODE=2 $B --workload tj_hard
IC3_PS_ZMODE=1 timeout 600 python -m pytest tests/test_policy_step_gpu.py -q -p no:cacheprovider -k "equals_the_launch_chain" > $O/tests_sc1.log 2>&1
IC3_PS_ZMODE=2 timeout 600 python -m pytest tests/test_policy_step_gpu.py -q -p no:cacheprovider -k "equals_the_launch_chain" > $O/tests_nt.log 2>&1
tail -n 1 $O/tests_sc1.log $O/tests_nt.log
