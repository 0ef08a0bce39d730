#!/bin/bash
# round 3, call B: GPU tests of the builtin-VMEM policy_step kernel + pacing sweep against the round-2 library
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 6 $O/pytest.log
L=$PWD/ic3net_amd/csrc
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-30s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run warm $B
run r02_1 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run new_default_1 $B
for zs in 5 6 7 8; do run zs$zs IC3_PS_ZS=$zs $B; done
for zs in 5 6; do for zf in 8 16 32; do run zs${zs}_zf$zf IC3_PS_ZS=$zs IC3_PS_ZF=$zf $B; done; done
for zs in 5 6; do for zh in 16 32; do run zs${zs}_zh$zh IC3_PS_ZS=$zs IC3_PS_ZH=$zh $B; done; done
run zs5_zf16_zh16 IC3_PS_ZS=5 IC3_PS_ZF=16 IC3_PS_ZH=16 $B
run zs6_zepi0 IC3_PS_ZS=6 IC3_PS_ZEPI=0 $B
run zs7_zepi0 IC3_PS_ZS=7 IC3_PS_ZEPI=0 $B
run r02_2 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run new_default_2 $B
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload tj_hard"
run tj_hard_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run tj_hard_new $B
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload tj_medium"
run tj_medium_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run tj_medium_new $B
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --no-dense-obs"
run noobs_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run noobs_new $B
