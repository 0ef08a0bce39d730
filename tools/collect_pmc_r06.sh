# PMC passes of an update (round 6): separate --pmc runs, counters only (no trace domains besides the kernel trace the tool needs)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pmc; mkdir -p $O
CMD="python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -o p -- $CMD > /dev/null 2>&1
for c in FETCH_SIZE; do python tools/collect_pmc.py $O/fetch $c | head -8; done
python tools/collect_pmc.py $O/write WRITE_SIZE | head -8
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do python tools/collect_pmc.py $O/sq $c | head -7; done
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES; do python tools/collect_pmc.py $O/sq2 $c | head -7; done
