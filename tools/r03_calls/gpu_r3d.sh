#!/bin/bash
# round 3, call D: first-round stagger + store slots in the front phases (pacing sweep, PP-hard)
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 600 python -m pytest tests/test_policy_step_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 2 $O/pytest.log
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run warm $B
run r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run base $B
for st in 4 8 12 16 20; do run stagger$st IC3_PS_STAGGER=$st $B; done
for st in 8 14; do run stagger${st}_zs5_zf16_zh16 IC3_PS_STAGGER=$st IC3_PS_ZS=5 IC3_PS_ZF=16 IC3_PS_ZH=16 $B; done
run zs4_z0_16_z3_16_zc16_zf16 IC3_PS_ZS=4 IC3_PS_Z0=16 IC3_PS_Z3=16 IC3_PS_ZC=16 IC3_PS_ZF=16 $B
run zs4_z3_16_zc16_zf16_zh16 IC3_PS_ZS=4 IC3_PS_Z3=16 IC3_PS_ZC=16 IC3_PS_ZF=16 IC3_PS_ZH=16 $B
run zs5_zc16_zf16 IC3_PS_ZS=5 IC3_PS_ZC=16 IC3_PS_ZF=16 $B
run zs5_z3_16_zf16 IC3_PS_ZS=5 IC3_PS_Z3=16 IC3_PS_ZF=16 $B
run zs5_z0_16_zf16 IC3_PS_ZS=5 IC3_PS_Z0=16 IC3_PS_ZF=16 $B
run zs4_z0_8_z3_16_zc16_zf16_zh16 IC3_PS_ZS=4 IC3_PS_Z0=8 IC3_PS_Z3=16 IC3_PS_ZC=16 IC3_PS_ZF=16 IC3_PS_ZH=16 $B
run st12_zs4_z3_16_zc16_zf16_zh16 IC3_PS_STAGGER=12 IC3_PS_ZS=4 IC3_PS_Z3=16 IC3_PS_ZC=16 IC3_PS_ZF=16 IC3_PS_ZH=16 $B
run r02_b IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run base_b $B
