#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3zf
mkdir -p $O
L=$PWD/ic3net_amd/csrc/libic3rollout_splitagpr.so
IC3_ROLLOUT_LIB=$L timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_step_onehop_gpu.py -m gpu -x -q -p no:cacheprovider -k "split" > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
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
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W --gate-split 1"
  run warm_$W $B
  run ${W}_split $B
  run ${W}_split_agpr IC3_ROLLOUT_LIB=$L $B
  run ${W}_split_b $B
  run ${W}_split_agpr_b IC3_ROLLOUT_LIB=$L $B
done
run pp_hard_noobs_split python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --gate-split 1 --no-dense-obs
run pp_hard_noobs_split_agpr IC3_ROLLOUT_LIB=$L python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --gate-split 1 --no-dense-obs
