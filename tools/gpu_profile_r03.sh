#!/bin/bash
# Round-3 evidence run: smoke, the whole GPU suite, the default bench command (+ driver arguments), its kernel trace /
# stats and the two HBM PMC passes (separate runs, as the microarch guide prescribes) for every BASELINE workload, SQ
# counters of the launch, the update half.  Output: gpurun_out/r03prof -> python tools/collect_profile_r03.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
rm -rf $O; mkdir -p $O
cd $R
./tools/exp/buf_probe > $O/buf_probe.txt 2>&1
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" > $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt
B="python bench.py --steps 160 --warmup 16 --no-cpu-baseline"
timeout 600 python bench.py --steps 160 --warmup 16 > $O/bench_pp_hard.json 2> $O/bench_pp_hard.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pp_hard_driver_args.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rccl 1 > $O/bench_pp_hard_driver_args_rccl_world1.json 2> $O/rccl_world1.err
for w in tj_hard tj_medium pp_easy; do timeout 300 $B --workload $w > $O/bench_$w.json 2> /dev/null; done
timeout 600 $B --workload pp_scaled --steps 24 --warmup 4 > $O/bench_pp_scaled.json 2> $O/bench_pp_scaled.err
timeout 300 $B --auto-reset 1 > $O/bench_pp_hard_auto_reset.json 2>/dev/null
timeout 300 $B --no-dense-obs > $O/bench_pp_hard_no_obs_diagnostic.json 2>/dev/null
timeout 300 $B --time-kernels 0 > $O/bench_pp_hard_graph.json 2>/dev/null
S="--steps 40 --warmup 8 --no-cpu-baseline"
for w in pp_hard tj_hard tj_medium; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python bench.py $S --workload $w > $O/kt_$w.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$w -- python bench.py $S --workload $w > $O/pmc_w_$w.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$w -- python bench.py $S --workload $w > $O/pmc_f_$w.log 2>&1
done
S="--steps 12 --warmup 3 --no-cpu-baseline --workload pp_scaled"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pp_scaled -- python bench.py $S > $O/kt_pp_scaled.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_pp_scaled -- python bench.py $S > $O/pmc_w_pp_scaled.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_pp_scaled -- python bench.py $S > $O/pmc_f_pp_scaled.log 2>&1
# SQ counters of the PP-hard launch, three per pass
S="--steps 20 --warmup 5 --no-cpu-baseline"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_$i -- python bench.py $S > $O/sq_$i.log 2>&1
done
timeout 900 python tools/bench_train.py 8192 3 native > $O/train_batch_8192_native.txt 2>&1
timeout 900 python tools/bench_train.py 1024 3 native > $O/train_batch_1024_native.txt 2>&1
timeout 900 python tools/bench_train.py 1024 3 autograd > $O/train_batch_1024_autograd.txt 2>&1
timeout 900 python tools/profile_train_native.py 8192 > $O/train_batch_8192_native_profile.txt 2>&1
IC3_ROLLOUT_LIB=$R/ic3net_amd/csrc/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace_pp_hard.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 > /dev/null 2>&1
python tools/analyze_trace.py $O/trace_pp_hard.csv > $O/phase_trace_pp_hard.txt 2>&1; rm -f $O/trace_pp_hard.csv
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cat $O/summary.txt; tail -n 2 $O/smoke.log; tail -n 1 $O/tests_gpu.log; du -sh $O
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys,os
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}
    print("%-46s %.4f ms/step %.1f M/s | hbm %.4f ms %.0f GB/s frac %.3f | mfma %s TF" % (os.path.basename(sys.argv[1]), d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), r.get('achieved',0), r.get('frac',0), m.get('achieved')))
except Exception as e: print(sys.argv[1],'FAILED',e)
PY
done
tail -n 1 $O/train_batch_*_native.txt $O/train_batch_1024_autograd.txt
