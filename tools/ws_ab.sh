#!/bin/bash
# A/B of the wave-specialised schedule (IC3_PS_WS=1) against the default kernel: tools/ws_ab.sh "workloads" [extra env for the ws leg]
export TMPDIR=/tmp
mkdir -p gpurun_out/ws
python bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1
for w in ${1:-pp_hard}; do for ws in 0 1; do
  env IC3_PS_WS=$ws $2 python bench.py --no-cpu-baseline --workload $w > gpurun_out/ws/${w}_$ws.json 2> gpurun_out/ws/${w}_$ws.err
  python - gpurun_out/ws/${w}_$ws.json "$w ws=$ws $2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-40s %.4f ms/step %7.1f M/s launch %.4f (min %.4f)" % (sys.argv[2], d["ms_per_step"], d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["timing"]["launch_ms_min"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-800:])
PY
done; done
