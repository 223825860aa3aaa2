#!/bin/bash
# round 3, run 1: full -m gpu suite (chunked quads, lane growth, max_angle, locate slack) + a short bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 --durations=20 > gpurun_out/r3_run1_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run1_tests.log
timeout 400 python bench.py --steps 100 --repeats 3 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register > gpurun_out/r3_run1_bench.json 2> gpurun_out/r3_run1_bench.err
echo "bench rc=$?" >> gpurun_out/r3_run1_tests.log
tail -40 gpurun_out/r3_run1_tests.log
