#!/bin/bash
# round 3, call C: operand ring 8 vs 4, TJ no-obs comparison, phase traces of the new kernel
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 600 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_gpu.py tests/test_policy_step_onehop_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
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
for W in pp_hard tj_medium tj_hard; do
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  run ${W}_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B
  run ${W}_ring8 $B
  run ${W}_ring4 IC3_ROLLOUT_LIB=$L/libic3rollout_ring4.so $B
  run ${W}_noobs_r02 IC3_ROLLOUT_LIB=$L/libic3rollout_r02.so $B --no-dense-obs
  run ${W}_noobs_ring8 $B --no-dense-obs
done
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run pp_hard_ring8_zf16_zh16 IC3_PS_ZS=5 IC3_PS_ZF=16 IC3_PS_ZH=16 $B
run pp_hard_ring8_zf32 IC3_PS_ZS=5 IC3_PS_ZF=32 $B
for W in pp_hard tj_medium; do
  IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace_$W.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 --workload $W > $O/trace_$W.json 2> $O/trace_$W.err
  python tools/analyze_trace.py $O/trace_$W.csv > $O/trace_$W.txt 2>&1
  head -n 24 $O/trace_$W.txt
  rm -f $O/trace_$W.csv
done
