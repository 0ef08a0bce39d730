#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | frac %.3f kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('frac',0), r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 24 --warmup 4 --workload pp_scaled"
run warm $B
run base $B
for s in 16 64 128 200; do run stagger$s IC3_PS_STAGGER=$s $B; done
run zs0 IC3_PS_ZS=0 $B
run zs8 IC3_PS_ZS=8 $B
run zf0_zc0 IC3_PS_ZF=0 IC3_PS_ZC=0 $B
run chain python bench.py --no-cpu-baseline --steps 24 --warmup 4 --workload pp_scaled --mega 0
run base_b $B
