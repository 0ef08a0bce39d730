#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
for D in 0 1 2 4 8 16 32 3 7 15 31 63; do IC3_PS_DEBUG=$D IC3_MB_OBS=1 timeout 200 python tools/microbench_policy_step.py pp_hard 384 48 2>/dev/null | grep "ic3_policy_step"; done | tee $O/lone_tile_ablation.txt
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
run warm            $B --steps 160 --warmup 16
run base            $B --steps 160 --warmup 16
run zl_1212         IC3_PS_ZL=0x1212 $B --steps 160 --warmup 16
run zl_1222         IC3_PS_ZL=0x1222 $B --steps 160 --warmup 16
run zl_1112         IC3_PS_ZL=0x1112 $B --steps 160 --warmup 16
run zl_2121         IC3_PS_ZL=0x2121 $B --steps 160 --warmup 16
run zl_0202         IC3_PS_ZL=0x0202 $B --steps 160 --warmup 16
run zl_1212_b       IC3_PS_ZL=0x1212 $B --steps 160 --warmup 16
run base_b          $B --steps 160 --warmup 16
