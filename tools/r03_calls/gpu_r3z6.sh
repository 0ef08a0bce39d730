#!/bin/bash
# round 3, call Z6: full GPU suite on the code with the gate_split experiment + its labelled bench lines
export TMPDIR=/tmp
O=gpurun_out/r3z6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_gpu.log 2>&1; tail -n 3 $O/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16"
for W in pp_hard tj_hard tj_medium pp_easy; do
  timeout 300 $B --workload $W > $O/bench_${W}_fp32_same_call.json 2>/dev/null
  timeout 300 $B --workload $W --gate-split 1 > $O/bench_${W}_EXPERIMENT_gate_split.json 2>/dev/null
done
timeout 300 $B --workload pp_hard --gate-split 1 --incremental-obs 1 > $O/bench_pp_hard_EXPERIMENT_gate_split_incremental_obs.json 2>/dev/null
timeout 300 $B --workload pp_hard --gate-split 1 --no-dense-obs > $O/bench_pp_hard_EXPERIMENT_gate_split_no_obs_diagnostic.json 2>/dev/null
timeout 300 $B --workload pp_scaled --steps 24 --warmup 4 --gate-split 1 > $O/bench_pp_scaled_EXPERIMENT_gate_split.json 2>/dev/null
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}; t=d.get('timing') or {}
    print("%-62s %.4f ms/step %6.1f M/s | launch %.4f (med %s) | mfma %s TF" % (os.path.basename(sys.argv[1]), d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_median'), m.get('achieved')))
except Exception as e: print(sys.argv[1],'FAILED',e)
PY
done
