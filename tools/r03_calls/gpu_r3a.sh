#!/bin/bash
# round 3, call A: range-check probe, GPU tests of the rewritten policy_step kernel, A/B against the round-2 library
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
./tools/exp/buf_probe > $O/buf_probe.txt 2>&1
cat $O/buf_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
L=$PWD/ic3net_amd/csrc
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print("%-28s %.4f ms/step  %.1f M/s | kernel %.4f ms %.0f GB/s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0)))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for W in pp_hard tj_hard; do
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  for rep in 1 2; do
    run ${W}_r02_$rep IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
    run ${W}_new_$rep $B
  done
done
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
for zs in 0 2 3 4 6 8; do run pp_hard_zs$zs IC3_PS_ZS=$zs $B; done
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard --no-dense-obs"
run pp_hard_noobs_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
run pp_hard_noobs_new $B
