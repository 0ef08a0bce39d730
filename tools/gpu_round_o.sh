#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02o
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
run nt            $B
run nt_st         IC3_PS_ZMODE=6 $B
run nt_ld         IC3_PS_ZMODE=10 $B
run nt_st_ld      IC3_PS_ZMODE=14 $B
run nt_zl2_zb0    IC3_PS_ZB=0 IC3_PS_ZL=2 $B
run nt_zb30       IC3_PS_ZB=30 $B
run nt_graph      $B --time-kernels 0
run plain         IC3_PS_ZMODE=0 $B
run tj_hard       $B --workload tj_hard
run tj_medium     $B --workload tj_medium
run pp_easy       $B --workload pp_easy
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_auto_reset_gpu.py -q -p no:cacheprovider > $O/tests.log 2>&1
tail -n 1 $O/tests.log
