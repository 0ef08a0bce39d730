#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/tests_ps.log 2>&1
echo "policy_step rc=$?" > $O/summary.txt
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}
    print("%-28s %.4f ms/step  %.1f M/s | hbm-kernel %.4f ms %.0f GB/s | step-kernel %s ms %s TF | %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0), m.get('avg_launch_ms'), m.get('achieved'), d['config']['launch']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
run fused_eager        $B
run fused_graph        $B --time-kernels 0
run separate_eager     $B --fused-obs 0
run tj_hard_fused      $B --workload tj_hard
run tj_hard_separate   $B --workload tj_hard --fused-obs 0
timeout 200 python tools/microbench_policy_step.py pp_hard 8192 40 mega >> $O/micro.txt 2>&1
grep -E "passed|failed|FAILED|Error" $O/tests_ps.log | tail -n 12; cat $O/summary.txt; grep median $O/micro.txt
tail -n 2 $O/*.err | grep -v amdgpu.ids | grep -v "^$" | grep -v "==>" | head
