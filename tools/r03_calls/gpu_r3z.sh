#!/bin/bash
# round 3, call Z: EXPERIMENT gate_split (exact bf16 split products in the gate GEMM of ic3_policy_step)
export TMPDIR=/tmp
O=gpurun_out/r3z
mkdir -p $O
timeout 900 python -m pytest tests/test_policy_step_gpu.py -m gpu -x -q -p no:cacheprovider -k "equals_the_launch_chain" > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
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
for W in pp_hard tj_hard tj_medium; do
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  run ${W}_fp32 $B
  run ${W}_split $B --gate-split 1
  run ${W}_fp32_b $B
  run ${W}_split_b $B --gate-split 1
done
run pp_hard_split_noobs python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --no-dense-obs --gate-split 1
run pp_scaled_split python bench.py --no-cpu-baseline --steps 24 --warmup 4 --workload pp_scaled --gate-split 1
