#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3z4
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --gate-split 1"
run warm $B
run base $B
for z in 2 3 4 6 8 10 12 16; do run zs$z IC3_PS_ZS=$z $B; done
run zs8_zepi0 IC3_PS_ZS=8 IC3_PS_ZEPI=0 $B
run zs6_zf0_zc0 IC3_PS_ZS=6 IC3_PS_ZF=0 IC3_PS_ZC=0 $B
run zs4_zf32_zc16 IC3_PS_ZS=4 IC3_PS_ZF=32 IC3_PS_ZC=16 $B
run zs6_z0_16_z3_16 IC3_PS_ZS=6 IC3_PS_Z0=16 IC3_PS_Z3=16 $B
run zs0 IC3_PS_ZS=0 $B
run base_b $B
