cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > $O/tests_gpu_summary.txt
( for w in tj_medium_commnet_mlp pp_hard_ic pp_hard_iric pp_hard_iric_tanh tj_hard tj_medium pp_scaled; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1; done ) > $O/bench_other_workloads.jsonl
bash tools/collect_pmc_commnet_r06.sh > $O/pmc_commnet_raw.txt 2>/dev/null
rm -rf gpurun_out/r06pmc2
cat $O/tests_gpu_summary.txt
