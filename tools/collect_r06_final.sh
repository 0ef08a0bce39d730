# Round 6, final code: everything profiles/r06 cites, in ONE gpurun call (tools/collect_r06.sh + the GPU suite + the PMC passes of the
# rollout launch and of commnet_forward_kernel; counters in runs of their own, kernel trace only).
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/tests_gpu_summary.txt
bash tools/collect_r06.sh > $O/collect.log 2>&1
# PMC: HBM traffic of the step launch (separate passes; FETCH_SIZE x 2 on gfx950 as the guide prescribes -> bench.py's from_profiles)
CMD="python bench.py --steps 40 --warmup 8 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_rollout.f -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_rollout.w -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_rollout.sq -o p -- $CMD > /dev/null 2>&1
( echo "# rocprofv3 --pmc passes of '$CMD' (one counter group per run); averages per launch, FETCH_SIZE / WRITE_SIZE in KiB as reported"
  python tools/collect_pmc.py $O/pmc_rollout.f FETCH_SIZE | sed -n 1,3p
  python tools/collect_pmc.py $O/pmc_rollout.w WRITE_SIZE | sed -n 2,3p
  for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do python tools/collect_pmc.py $O/pmc_rollout.sq $c | sed -n 2p; done ) > $O/pmc_rollout.txt
bash tools/collect_pmc_commnet_r06.sh > $O/pmc_commnet_raw.txt 2>/dev/null
rm -rf $O/pmc_rollout.f $O/pmc_rollout.w $O/pmc_rollout.sq gpurun_out/r06pmc2/*.sq gpurun_out/r06pmc2/*.sq2 gpurun_out/r06pmc2/*.f gpurun_out/r06pmc2/*.w
cp $O/prof_rollout/*kernel_stats.csv $O/rollout_kernel_stats.csv 2>/dev/null
cp $O/prof_train/*kernel_stats.csv $O/train_kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +2000k -delete; find $O -name "*trace*.csv" -delete
ls -la $O
cat $O/tests_gpu_summary.txt $O/pmc_rollout.txt
