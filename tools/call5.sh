export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace
mkdir -p $O
cd /tmp
for cfg in "IC3_FILL_WAVES=400 IC3_FILL_NAP=0" "IC3_FILL_WAVES=400 IC3_FILL_NAP=6" "IC3_FILL_WAVES=100 IC3_FILL_NAP=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --prefill-obs 1 --gate-split 1 --steps 40 --warmup 8 > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_trace.csv" | head -1)
  python - "$f" "$tag" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'policy_step_kernel' in r['Kernel_Name'] or 'obs_fill_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-24:]
t0=int(rows[0]['Start_Timestamp'])
print("==", sys.argv[2])
for r in rows:
    print("%-18s start %9.1f us  end %9.1f us  dur %7.1f  q %s" % (r['Kernel_Name'][:18], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?')))
PY
  rm -rf $O/$tag
done
