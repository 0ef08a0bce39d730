#!/bin/bash
# round 3, call F: full GPU suite on the current code, TJ-hard phase traces (obs / no obs), native update throughput + profile
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
L=$PWD/ic3net_amd/csrc
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1
tail -n 6 $O/pytest.log
run() { name=$1; shift; timeout 300 env "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    t=d.get('timing') or {}
    print("%-34s %.4f ms/step  %.1f M/s | kernel avg %.4f min %s med %s" % (sys.argv[2], d['ms_per_step'], d['value']/1e6, r.get('avg_launch_ms',0), t.get('launch_ms_min'), t.get('launch_ms_median')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload tj_hard"
run warm_tj_hard $B
run tj_hard_default $B
run tj_hard_zs3_zepi0 IC3_PS_ZS=3 IC3_PS_ZEPI=0 IC3_PS_ZC=0 IC3_PS_ZF=0 $B
run tj_hard_zs2_zf14 IC3_PS_ZS=2 IC3_PS_ZEPI=0 IC3_PS_ZC=0 IC3_PS_ZF=14 $B
run tj_hard_zs0 IC3_PS_ZS=0 IC3_PS_ZEPI=0 IC3_PS_ZC=0 IC3_PS_ZF=0 $B
for V in obs noobs; do
  X=""; [ $V = noobs ] && X="--no-dense-obs"
  IC3_ROLLOUT_LIB=$L/libic3rollout_trace.so IC3_PS_TRACE_OUT=$O/trace_tj_hard_$V.csv timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 16 --workload tj_hard $X > $O/trace_tj_hard_$V.json 2> $O/trace_tj_hard_$V.err
  python tools/analyze_trace.py $O/trace_tj_hard_$V.csv > $O/trace_tj_hard_$V.txt 2>&1
  sed -n 1,24p $O/trace_tj_hard_$V.txt
  rm -f $O/trace_tj_hard_$V.csv
done
B="python bench.py --no-cpu-baseline --steps 160 --warmup 16 --workload pp_hard"
run pp_hard_default $B
timeout 900 python tools/bench_train.py 8192 3 native > $O/train_8192_native.txt 2>&1; tail -n 1 $O/train_8192_native.txt
timeout 900 python tools/profile_train_native.py 8192 > $O/train_8192_native_profile.txt 2>&1; tail -n 40 $O/train_8192_native_profile.txt
