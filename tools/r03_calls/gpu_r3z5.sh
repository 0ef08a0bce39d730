#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3z
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_onehop_gpu.py -m gpu -x -q -p no:cacheprovider -k "gate_split" > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
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
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run pp_hard_incr $B --incremental-obs 1
run pp_hard_incr_split $B --incremental-obs 1 --gate-split 1
run pp_easy_split python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_easy --gate-split 1
run pp_easy_fp32 python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_easy
