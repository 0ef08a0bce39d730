set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python bench.py > $O/bench_rollout.json 2> $O/bench_rollout.err
python bench.py --mode train > $O/bench_train.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rollout -o rollout -- python bench.py --no-cpu-baseline > $O/prof_rollout.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python bench.py --mode train --no-cpu-baseline > $O/prof_train.json 2>/dev/null
( for w in pp_hard tj_hard tj_medium; do python tools/bench_train.py 8192 6 native $w 2>&1 | tail -1; done
  python tools/bench_train.py 8192 6 native pp_hard 1 1 1 2>&1 | tail -1
  python tools/bench_train.py 8192 3 native pp_scaled 2>&1 | tail -1
  TWO_CHAINS=0 python tools/bench_train.py 8192 6 native pp_hard 2>&1 | tail -1
  TWO_CHAINS=0 ENC_WINDOW=0 python tools/bench_train.py 8192 6 native pp_hard 2>&1 | tail -1 ) > $O/train_batch.txt 2>&1
( for w in tj_medium_commnet_mlp pp_hard_ic pp_hard_iric pp_hard_iric_tanh tj_hard tj_medium pp_scaled; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done ) > $O/bench_other_workloads.jsonl
python tools/microbench_bptt.py > $O/microbench_bptt.txt 2>&1
python tools/bench_collection.py > $O/collection.txt 2>&1
ls $O $O/prof_rollout $O/prof_train
