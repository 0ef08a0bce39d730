#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02aj
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
for zb in 0 2 4 6 8 12 16; do run zb$zb IC3_PS_ZB=$zb $B; done
for zb in 4 8 12; do run zb${zb}_f60 IC3_PS_ZB=$zb IC3_PS_ZFRAC=60 $B; done
for zb in 4 8 12; do run zb${zb}_f50 IC3_PS_ZB=$zb IC3_PS_ZFRAC=50 $B; done
run zb8_tjh IC3_PS_ZB=8 $B --workload tj_hard
run zb0_tjh IC3_PS_ZB=0 $B --workload tj_hard
