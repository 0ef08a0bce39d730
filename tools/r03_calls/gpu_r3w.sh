#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03prof
mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rccl 1 > $O/bench_pp_hard_driver_args_rccl_world1.json 2> $O/rccl_world1.err; echo rc=$?
tail -c 600 $O/bench_pp_hard_driver_args_rccl_world1.json; tail -n 15 $O/rccl_world1.err
