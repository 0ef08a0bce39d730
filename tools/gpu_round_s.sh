#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02s
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
B="python bench.py --no-cpu-baseline"
timeout 600 python -m pytest tests/test_policy_step_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -n 1 $O/tests.log
run warm            $B --steps 160 --warmup 16
run base            $B --steps 160 --warmup 16
run nowarmup_c      IC3_PS_ZMODE=18 $B --steps 160 --warmup 16
run norefill        IC3_PS_DEBUG=64 $B --steps 160 --warmup 16
run zl_0004         IC3_PS_ZL=0x0004 $B --steps 160 --warmup 16
run zl_0022         IC3_PS_ZL=0x0022 $B --steps 160 --warmup 16
run zl_0013         IC3_PS_ZL=0x0013 $B --steps 160 --warmup 16
run zl_2222         IC3_PS_ZL=0x2222 $B --steps 160 --warmup 16
run zl_0044         IC3_PS_ZL=0x0044 $B --steps 160 --warmup 16
run zl_0008         IC3_PS_ZL=0x0008 $B --steps 160 --warmup 16
run zl_1111_b       $B --steps 160 --warmup 16
run s20_w5          $B --steps 20 --warmup 5
