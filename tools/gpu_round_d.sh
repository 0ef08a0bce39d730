#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}
    print("%-34s %.4f ms/step  %.1f M/s live %.3f | obs %.4f ms | step-kernel %s ms | %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, d.get('live_frac',0), r.get('avg_launch_ms',0), m.get('avg_launch_ms'), d['config']['launch']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
run base_eager           $B
run base_graph           $B --time-kernels 0
run overlap_eager        $B --overlap-obs 1
run overlap_graph        $B --overlap-obs 1 --time-kernels 0
run wg1_eager            IC3_PS_WGS=1 $B
run wg1_overlap_eager    IC3_PS_WGS=1 $B --overlap-obs 1
run wg1_overlap_graph    IC3_PS_WGS=1 $B --overlap-obs 1 --time-kernels 0
run chain_overlap_graph  $B --mega 0 --overlap-obs 1 --time-kernels 0
tail -n 3 $O/*.err | grep -v amdgpu.ids | grep -v "^$" | head -20
