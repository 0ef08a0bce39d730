#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_policy_step_onehop_gpu.py tests/test_auto_reset_gpu.py tests/test_policy_gpu.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
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
  B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload $W"
  run warm_$W $B
  run ${W}_s0 $B
  run ${W}_s0_enc4 IC3_ROLLOUT_LIB=$L/libic3rollout_enc4.so $B
  run ${W}_s0_b $B
  run ${W}_noobs $B --no-dense-obs
done
IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 > /dev/null 2>&1
python tools/analyze_trace.py $O/trace.csv > $O/trace_pp_hard.txt 2>&1; sed -n 1,24p $O/trace_pp_hard.txt; rm -f $O/trace.csv
