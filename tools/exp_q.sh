cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_commnet_step_gpu.py tests/test_trainer_gpu.py tests/test_auto_reset_gpu.py tests/test_main_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python tools/soak_obs_rows.py 30 2>&1 | grep -v amdgpu.ids | head -3
for w in tj_medium_commnet_mlp pp_hard_ic pp_hard_iric_tanh; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/line_$w.json; python -c "
import json; d=json.loads(open('gpurun_out/line_$w.json').read()); r=d['roofline']; print('$w', round(d['value']/1e6,1), d['ms_per_step'], r['avg_launch_ms'], r['frac'])"; done
