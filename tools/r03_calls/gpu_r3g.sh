#!/bin/bash
# round 3, call G: one gate loop at a time per CU (IC3_PS_GATELOCK) on/off, traces
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_step_onehop_gpu.py tests/test_auto_reset_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for W in pp_hard tj_hard tj_medium pp_easy; do
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  run ${W}_lock1 $B
  run ${W}_lock0 IC3_PS_GATELOCK=0 $B
  run ${W}_noobs_lock1 $B --no-dense-obs
  run ${W}_noobs_lock0 IC3_PS_GATELOCK=0 $B --no-dense-obs
done
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run pp_hard_lock1_zs7 IC3_PS_ZS=7 IC3_PS_ZC=0 IC3_PS_ZF=0 IC3_PS_ZEPI=0 $B
run pp_hard_lock1_zs6 IC3_PS_ZS=6 $B
run pp_hard_lock1_zs4 IC3_PS_ZS=4 $B
for W in pp_hard; do
  IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace_$W.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 --workload $W > $O/trace_$W.json 2> $O/trace_$W.err
  python tools/analyze_trace.py $O/trace_$W.csv > $O/trace_$W.txt 2>&1
  sed -n 1,46p $O/trace_$W.txt; tail -n 4 $O/trace_$W.txt
  rm -f $O/trace_$W.csv
done
