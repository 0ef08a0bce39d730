#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ak
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
run base $B
for w in tj_hard tj_medium pp_easy; do run ${w} $B --workload $w; for zb in 0 1 2 4; do run ${w}_zb$zb IC3_PS_ZB=$zb $B --workload $w; done; done
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_trainer_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -n 1
