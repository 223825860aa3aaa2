#!/bin/bash
# round 3, run 13: the whole GPU suite, then the final pass (tools/r3_run12.sh)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > gpurun_out/r3_run13_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run13_tests.log
tail -25 gpurun_out/r3_run13_tests.log
bash tools/r3_run12.sh
