#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3z7
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=[json.loads(l) for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-40s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --gate-split 1"
run warm $B
run base $B
run zs3_z0_32_z3_32_zf48_zc16 IC3_PS_ZS=3 IC3_PS_Z0=32 IC3_PS_Z3=32 IC3_PS_ZF=48 IC3_PS_ZC=16 $B
run zs4_z0_16_z3_16_zf32_zc16 IC3_PS_ZS=4 IC3_PS_Z0=16 IC3_PS_Z3=16 IC3_PS_ZF=32 IC3_PS_ZC=16 $B
run zs4_z0_32_zf32_zc16 IC3_PS_ZS=4 IC3_PS_Z0=32 IC3_PS_ZF=32 IC3_PS_ZC=16 $B
run zs4_zf48_zc16_zh16 IC3_PS_ZS=4 IC3_PS_ZF=48 IC3_PS_ZC=16 IC3_PS_ZH=16 $B
run zs4_zf32_zc16_zh32 IC3_PS_ZS=4 IC3_PS_ZF=32 IC3_PS_ZC=16 IC3_PS_ZH=32 $B
run zs5_zf32_zc16 IC3_PS_ZS=5 IC3_PS_ZF=32 IC3_PS_ZC=16 $B
run zs3_zf64_zc16 IC3_PS_ZS=3 IC3_PS_ZF=64 IC3_PS_ZC=16 $B
run zs4_zf32_zc16_zepi1 IC3_PS_ZS=4 IC3_PS_ZF=32 IC3_PS_ZC=16 IC3_PS_ZEPI=1 $B
run base_b $B
