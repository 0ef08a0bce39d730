#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02ah
mkdir -p $O
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
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16"
timeout 1200 python -m pytest tests/test_episode_finalize_gpu.py tests/test_trainer_gpu.py tests/test_auto_reset_gpu.py tests/test_env_parity_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -n 2 $O/tests.log
run warm    $B
run base_1  $B
run base_2  $B
run graph   $B --time-kernels 0
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $GRAFT_REPO_ROOT/$O/kt.log 2>&1; cd $GRAFT_REPO_ROOT
head -12 $O/kt/*/*kernel_stats.csv | cut -c1-150
