#!/bin/bash
# Round-2 evidence run: smoke, the whole GPU suite, the default bench command with its kernel trace / stats and the two
# HBM PMC passes (separate runs, as the microarch guide prescribes), the other workloads, the separate / chain variants.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
rm -rf $O; mkdir -p $O
cd $R
make -s -C ic3net_amd/csrc libic3rollout_plain.so > /dev/null 2>&1    # A/B build of the same sources (never stale)
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
timeout 600 python bench.py --steps 160 --warmup 16 > $O/bench_pp_hard.json 2> $O/bench_pp_hard.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pp_hard_driver_args.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- $B > $O/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- $B > $O/pmc_f.log 2>&1
for w in tj_hard tj_medium pp_easy; do timeout 300 $B --workload $w > $O/bench_$w.json 2> /dev/null; done
timeout 300 $B --workload pp_scaled --nenvs 2048 --steps 20 --warmup 4 > $O/bench_pp_scaled_e2048.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_tj_hard -- $B --workload tj_hard > $O/kt_tj_hard.log 2>&1
timeout 300 $B --fused-obs 0 > $O/bench_pp_hard_separate_obs.json 2>/dev/null
timeout 300 $B --fused-obs 0 --overlap-obs 1 --time-kernels 0 > $O/bench_pp_hard_overlap_obs.json 2>/dev/null
timeout 300 $B --mega 0 --time-kernels 0 > $O/bench_pp_hard_chain_r01.json 2>/dev/null
timeout 300 $B --time-kernels 0 > $O/bench_pp_hard_graph.json 2>/dev/null
timeout 300 $B --auto-reset 1 > $O/bench_pp_hard_auto_reset.json 2>/dev/null
IC3_ROLLOUT_LIB=$R/ic3net_amd/csrc/libic3rollout_plain.so timeout 300 $B > $O/bench_pp_hard_plain_stores.json 2>/dev/null
IC3_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --output-format csv -d $O/roctx -- python bench.py --steps 8 --warmup 2 --nenvs 1024 --no-cpu-baseline > $O/roctx.log 2>&1
./tools/exp/ws_probe > $O/ws_probe.txt 2>&1
timeout 600 python tools/exp/two_stream.py pp_hard 8192 2>&1 | grep -v amdgpu.ids > $O/two_stream.txt
for D in 0 1 2 4 8 16 32; do IC3_PS_DEBUG=$D IC3_MB_OBS=1 timeout 200 python tools/microbench_policy_step.py pp_hard 8192 48 2>/dev/null | grep "ic3_policy_step"; done > $O/policy_step_ablation_e8192.txt
for D in 0 1 2 4 8 16 32; do IC3_PS_DEBUG=$D IC3_MB_OBS=1 timeout 200 python tools/microbench_policy_step.py pp_hard 384 48 2>/dev/null | grep "ic3_policy_step"; done > $O/policy_step_ablation_lone_tile_e384.txt
for E in 384 3072 6144 8192 9216 12288; do for OB in 0 1; do IC3_MB_OBS=$OB timeout 200 python tools/microbench_policy_step.py pp_hard $E 48 2>/dev/null | grep "ic3_policy_step" | sed "s/^/obs=$OB /"; done; done > $O/policy_step_env_count_sweep.txt
cat $O/summary.txt; tail -n 2 $O/smoke.log; tail -n 1 $O/tests_gpu.log; du -sh $O
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys,os
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}
    print("%-40s %.4f ms/step %.1f M/s | hbm %.4f ms %.0f GB/s frac %.3f | mfma %s TF" % (os.path.basename(sys.argv[1]), d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0), r.get('frac',0), m.get('achieved')))
except Exception as e: print(sys.argv[1],'FAILED',e)
PY
done
