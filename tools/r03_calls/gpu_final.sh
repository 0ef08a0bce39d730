#!/bin/bash
# last call of round 3: the whole GPU suite + smoke + the default bench line on the final library
# (after the ic3_env_observe_at snapshot fix and the IC3_DYNAMIC_LDS spelling)
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 240 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/tests_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -n 3 $O/tests_gpu.log
timeout 60 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/smoke.log
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench.err; tail -c 600 $O/bench_driver_args.json
