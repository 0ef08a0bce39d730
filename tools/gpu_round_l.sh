#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
timeout 900 python -m pytest tests/test_auto_reset_gpu.py tests/test_multirank_gpu.py -q --maxfail=30 -p no:cacheprovider > $O/tests_new.log 2>&1
echo "new tests rc=$?" > $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --deselect tests/test_auto_reset_gpu.py --deselect tests/test_multirank_gpu.py > $O/tests_all.log 2>&1
echo "suite rc=$?" >> $O/summary.txt
grep -E "passed|failed|FAILED|Error" $O/tests_new.log | tail -n 12; grep -E "passed|failed|FAILED" $O/tests_all.log | tail -n 8; cat $O/summary.txt
