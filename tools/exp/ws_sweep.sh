export TMPDIR=/tmp
python bench.py --no-cpu-baseline --steps 40 > /dev/null 2>&1
run() { env IC3_PS_WS=1 $1 python bench.py --no-cpu-baseline --workload ${2:-pp_hard} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %-10s %.4f ms/step  launch %.4f (min %.4f)' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['timing']['launch_ms_min']))" "$1" "${2:-pp_hard}"; }
for zs in 0 4 8 16; do for ze in 0 2; do run "IC3_WS_ZS=$zs IC3_WS_ZEPI=$ze"; done; done
run "IC3_WS_ZS=4 IC3_WS_ZEPI=2" tj_hard; run "IC3_WS_ZS=0 IC3_WS_ZEPI=2" tj_hard; run "IC3_WS_ZS=0 IC3_WS_ZEPI=0" tj_medium; run "IC3_WS_ZS=4 IC3_WS_ZEPI=0" tj_medium
