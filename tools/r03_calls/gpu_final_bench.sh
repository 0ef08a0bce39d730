#!/bin/bash
# the default bench command (+ TJ-hard, TJ-medium) on the last library of round 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_bench; rm -rf $O; mkdir -p $O
timeout 150 python bench.py > $O/bench_pp_hard_default_final_library.json 2> $O/bench.err
timeout 60 python bench.py --workload tj_hard --no-cpu-baseline > $O/bench_tj_hard_final_library.json 2>/dev/null
timeout 60 python bench.py --workload tj_medium --no-cpu-baseline > $O/bench_tj_medium_final_library.json 2>/dev/null
for f in $O/*.json; do python - $f <<'PY'
import json,sys,os
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print("%-50s %.4f ms/step %.1f M/s | launch %.4f ms frac %.3f" % (os.path.basename(sys.argv[1]), d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('frac',0)))
PY
done
