#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02ad
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $O/counters.txt
cd $R
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- bash -c "cd $R && $B" > $O/pmc$i.log 2>&1)
done
python - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r02ad'
for f in sorted(glob.glob(O+'/pmc*/*/*_counter_collection.csv')):
    d=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if 'policy_step_kernel' not in r['Kernel_Name']: continue
        d[r['Counter_Name']][0]+=1; d[r['Counter_Name']][1]+=float(r['Counter_Value'])
    for k,(n,v) in d.items(): print("%-28s %16.0f per launch (%d launches)"%(k,v/n,n))
PY
