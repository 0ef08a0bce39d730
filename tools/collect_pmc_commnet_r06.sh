# SQ / TCC counters of commnet_forward_kernel (round-5 verdict: none had been collected) and the rocprofv3 kernel average of PP-scaled
# beside its event timing; separate --pmc runs, counters only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pmc2; mkdir -p $O
for W in tj_medium_commnet_mlp pp_hard_ic; do
  CMD="python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/$W.sq -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/$W.sq2 -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$W.f -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$W.w -o p -- $CMD > /dev/null 2>&1
  echo "== $W"
  for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do python tools/collect_pmc.py $O/$W.sq $c | sed -n 2p; done
  for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES; do python tools/collect_pmc.py $O/$W.sq2 $c | sed -n 2p; done
  python tools/collect_pmc.py $O/$W.f FETCH_SIZE | sed -n 2p
  python tools/collect_pmc.py $O/$W.w WRITE_SIZE | sed -n 2p
done
echo "== pp_scaled: event-timed launch of the bench line against the rocprofv3 kernel average of the same command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/scaled -o s -- python bench.py --workload pp_scaled --no-cpu-baseline > $O/scaled.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/scaled.json').read().strip().splitlines()[-1]); print('bench line: ms_per_step', d['ms_per_step'], 'event-timed launch', d['roofline']['avg_launch_ms'])"
head -2 $O/scaled/s_kernel_stats.csv | cut -c1-160
