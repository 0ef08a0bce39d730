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
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}; m=d.get('roofline_mfma') or {}; t=d.get('timing') or {}
    print("%-44s %.4f ms/step %7.1f M/s | launch %.4f (min %s med %s) fill %s | hbm %.3f mfma %.3f | ok=%s" % (
        sys.argv[2], d['ms_per_step'], d['value']/1e6, (m or r).get('avg_launch_ms',0), t.get('launch_ms_min'),
        t.get('launch_ms_median'), d.get('fill_launch_ms'), r.get('frac',0) or 0, m.get('frac',0) or 0, t.get('consistent')))
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
sh)
  bash -c "$*" 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt | tail -40 ;;
esac
