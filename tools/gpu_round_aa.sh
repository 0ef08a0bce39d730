#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02aa
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
B="python bench.py --no-cpu-baseline"
timeout 900 python -m pytest tests/test_policy_step_gpu.py tests/test_trainer_gpu.py tests/test_auto_reset_gpu.py tests/test_env_parity_gpu.py tests/test_policy_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1
tail -n 3 $O/tests.log
run warm            $B --steps 160 --warmup 16
run base            $B --steps 160 --warmup 16
run nostage         IC3_PS_STAGE=0 $B --steps 160 --warmup 16
run base_b          $B --steps 160 --warmup 16
run nostage_b       IC3_PS_STAGE=0 $B --steps 160 --warmup 16
run s20_w5          $B --steps 20 --warmup 5
run tj_hard         $B --steps 160 --warmup 16 --workload tj_hard
run tj_hard_nostage IC3_PS_STAGE=0 $B --steps 160 --warmup 16 --workload tj_hard
run tj_medium       $B --steps 160 --warmup 16 --workload tj_medium
for D in 0 8; do IC3_PS_DEBUG=$D IC3_MB_OBS=1 timeout 200 python tools/microbench_policy_step.py pp_hard 384 48 2>/dev/null | grep "ic3_policy_step"; done
