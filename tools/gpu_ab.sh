#!/bin/bash
# A/B harness for one gpurun call: `VARIANTS="name1 name2" ENVS="IC3_PS_ZB=4;IC3_PS_ZFRAC=60" bash tools/gpu_ab.sh [workload]`
#   VARIANTS: builds made with tools/build_variant.sh (ic3net_amd/csrc/libic3rollout_<name>.so), run against the default
#   ENVS:     ';'-separated environment settings, each run against the default build
export TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O
W=${1:-pp_hard}
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
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
L=$PWD/ic3net_amd/csrc
run warm $B
for rep in 1 2; do
  run base_$rep $B
  for v in $VARIANTS; do run ${v}_$rep IC3_ROLLOUT_LIB=$L/libic3rollout_$v.so $B; done
  IFS=';' read -ra ES <<< "$ENVS"
  for e in "${ES[@]}"; do [ -n "$e" ] && run "$(echo $e | tr ' =' '__')_$rep" $e $B; done
done
for v in $VARIANTS; do
  IC3_ROLLOUT_LIB=$L/libic3rollout_$v.so timeout 600 python -m pytest tests/test_policy_step_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -n 1
done
