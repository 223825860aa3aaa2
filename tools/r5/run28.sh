#!/bin/bash
# round 5, run 28: the driver's own GPU test command, one process, on the shipped tree
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run28; mkdir -p $O
( time timeout -s KILL 640 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=5 ) > $O/gpu_tests_one_process.log 2>&1; echo "rc=$?" >> $O/gpu_tests_one_process.log
tail -14 $O/gpu_tests_one_process.log
