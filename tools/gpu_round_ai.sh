#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ai
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
run warm    $B
for f in 70 50 60 80 90 100 70; do run frac$f IC3_PS_ZFRAC=$f $B; done
run zc16 IC3_PS_ZC=16 $B
run zb4 IC3_PS_ZB=4 $B
