#!/bin/bash
# ONE parameterised script for the gpurun calls of a round (replaces round 3's tools/r03_calls/gpu_r3*.sh):
#   gpurun --timeout S -- 'bash tools/gpu_call.sh RECIPE [args...]'      -> gpurun_out/<RECIPE>/
# Recipes:
#   tests [pytest-args]        pytest -m gpu
#   bench NAME [bench-args]    one bench.py line -> NAME.json (+ .err)
#   matrix WORKLOADS -- "ARGS1" "ARGS2" ...    every workload x every argument set, one summary line each
#   probe PATTERN              tools/exp/ws_probe (configs whose name contains PATTERN)
#   prof NAME [bench-args]     rocprofv3 --kernel-trace --stats of a bench command; pmc NAME COUNTER [bench-args]
#   sh "command"               anything else, logged
export TMPDIR=/tmp
R=$1; shift
O=gpurun_out/$R
mkdir -p $O
summ() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}; t=d.get('timing') or {}
    print("%-44s %.4f ms/step %7.1f M/s | launch %.4f (min %s med %s) fill %s | hbm %.3f mfma %.3f | ok=%s" % (
        sys.argv[2], d['ms_per_step'], d['value']/1e6, (m or r).get('avg_launch_ms',0), t.get('launch_ms_min'),
        t.get('launch_ms_median'), None, r.get('frac',0) or 0, m.get('frac',0) or 0, t.get('consistent')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
    try: print(open(sys.argv[1][:-5]+'.err').read()[-1500:])
    except Exception: pass
PY
}
case $R in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" 2>&1 | grep -v amdgpu.ids | tee $O/pytest.txt | tail -15 ;;
bench)
  N=$1; shift
  timeout 600 python bench.py "$@" > $O/$N.json 2> $O/$N.err; summ $O/$N.json $N ;;
matrix)
  WL=$1; shift; [ "$1" == "--" ] && shift
  [ -z "$NOWARM" ] && timeout 300 python bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1    # warm the box
  for rep in $(seq 1 ${REPS:-1}); do for w in $WL; do i=0; for A in "$@"; do i=$((i+1))
    n=${w}_a${i}_r${rep}
    timeout 300 env $ENVPRE python bench.py --no-cpu-baseline --workload $w $A > $O/$n.json 2> $O/$n.err
    summ $O/$n.json "$w [$A] #$rep"
  done; done; done | tee $O/summary.txt ;;
probe)
  timeout 300 tools/exp/ws_probe "$@" 2>&1 | tee $O/ws_probe.txt ;;
prof)
  N=$1; shift
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$N -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$O/$N.json 2> $GRAFT_REPO_ROOT/$O/$N.err
  cd $GRAFT_REPO_ROOT; f=$(find $O/$N -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${N}_kernel_stats.csv && head -8 $O/${N}_kernel_stats.csv | cut -c1-200
  rm -rf $O/$N ;;
pmc)
  N=$1; C=$2; shift; shift
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/$N -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$O/$N.json 2> $GRAFT_REPO_ROOT/$O/$N.err
  cd $GRAFT_REPO_ROOT; python tools/collect_pmc.py $O/$N $C > $O/${N}_${C}.txt 2>&1; cat $O/${N}_${C}.txt | tail -12
  rm -rf $O/$N ;;
evidence)
  # the round's evidence in one call: bench lines of every BASELINE workload (default path), the fp32-instruction and
  # hipGraph variants of the headline, rocprofv3 kernel stats, PMC HBM passes and SQ counter passes -> gpurun_out/evidence/
  B="python bench.py --no-cpu-baseline"
  timeout 300 $B --steps 40 > /dev/null 2>&1
  timeout 600 python bench.py > $O/bench_pp_hard.json 2> $O/bench_pp_hard.err; summ $O/bench_pp_hard.json "pp_hard (default command)"
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_pp_hard_driver_args.json 2>/dev/null; summ $O/bench_pp_hard_driver_args.json "pp_hard --steps 20 --warmup 5"
  for w in tj_hard tj_medium pp_easy tj_medium_commnet_mlp pp_hard_ic pp_hard_iric; do timeout 300 $B --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; summ $O/bench_$w.json $w; done
  timeout 600 $B --workload pp_scaled --steps 40 > $O/bench_pp_scaled.json 2> $O/bench_pp_scaled.err; summ $O/bench_pp_scaled.json pp_scaled
  timeout 300 $B --gate-split 0 > $O/bench_pp_hard_fp32_instruction.json 2>/dev/null; summ $O/bench_pp_hard_fp32_instruction.json "pp_hard --gate-split 0"
  for w in tj_hard tj_medium; do timeout 300 $B --workload $w --gate-split 0 > $O/bench_${w}_fp32_instruction.json 2>/dev/null; summ $O/bench_${w}_fp32_instruction.json "$w --gate-split 0"; done
  timeout 300 $B --time-kernels 0 > $O/bench_pp_hard_graph.json 2>/dev/null; summ $O/bench_pp_hard_graph.json "pp_hard hipGraph replay"
  timeout 300 $B --no-dense-obs > $O/bench_pp_hard_no_obs_diagnostic.json 2>/dev/null; summ $O/bench_pp_hard_no_obs_diagnostic.json "pp_hard --no-dense-obs (diagnostic)"
  timeout 300 $B --auto-reset 1 > $O/bench_pp_hard_auto_reset.json 2>/dev/null; summ $O/bench_pp_hard_auto_reset.json "pp_hard --auto-reset 1"
  timeout 300 $B --rccl 1 --steps 20 --warmup 5 > $O/bench_pp_hard_rccl_world1.json 2>/dev/null; summ $O/bench_pp_hard_rccl_world1.json "pp_hard --rccl 1 (one-rank RCCL group)"
  cd /tmp
  S="--no-cpu-baseline --steps 40 --warmup 8"
  R=$GRAFT_REPO_ROOT
  for w in pp_hard tj_hard tj_medium pp_scaled tj_medium_commnet_mlp pp_hard_ic pp_hard_iric; do
    st="$S"; [ $w == pp_scaled ] && st="--no-cpu-baseline --steps 10 --warmup 4"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$w -- python $R/bench.py $st --workload $w > /dev/null 2>&1
    f=$(find $R/$O/kt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$O/bench_${w}_kernel_stats.csv
    for c in WRITE_SIZE FETCH_SIZE; do
      timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_${c}_$w -- python $R/bench.py $st --workload $w > /dev/null 2>&1
      python $R/tools/collect_pmc.py $R/$O/pmc_${c}_$w $c > $R/$O/pmc_${c}_$w.csv 2>&1; rm -rf $R/$O/pmc_${c}_$w
    done
    rm -rf $R/$O/kt_$w
  done
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/sq_$i -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
    for c in $grp; do python $R/tools/collect_pmc.py $R/$O/sq_$i $c | grep policy_step_kernel | head -1; done >> $R/$O/sq_counters.csv
    rm -rf $R/$O/sq_$i
  done
  cd $R; cat $O/sq_counters.csv; grep policy_step $O/pmc_*pp_hard.csv ;;
train)
  # the update half: tools/bench_train.py lines (lock-step and collection mode) + the kernel table of one PP-hard update
  for w in pp_hard tj_hard tj_medium tj_medium_commnet_mlp; do timeout 600 python tools/bench_train.py 8192 4 native $w 2>&1 | grep train_batch; done | tee $O/train_batch.txt
  timeout 600 python tools/bench_train.py 8192 4 native pp_hard 1 2>&1 | grep train_batch | tee -a $O/train_batch.txt
  timeout 600 python tools/bench_train.py 8192 4 native pp_hard 0 1 1 2>&1 | grep train_batch | tee -a $O/train_batch.txt
  timeout 600 python tools/bench_train.py 1024 4 native pp_hard 2>&1 | grep train_batch | tee -a $O/train_batch.txt
  timeout 600 python tools/bench_train.py 8192 2 native pp_hard_iric 2>&1 | grep train_batch | tee -a $O/train_batch.txt
  timeout 600 python tools/bench_train.py 8192 2 native pp_hard_ic 2>&1 | grep train_batch | tee -a $O/train_batch.txt
  timeout 600 python tools/profile_train_native.py 8192 2>&1 | grep -v "Warn\|amdgpu.ids\|_warn" > $O/train_profile.txt; tail -3 $O/train_profile.txt ;;
sh)
  bash -c "$*" 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt | tail -40 ;;
esac
