#!/bin/bash
# round 3, run 2: full -m gpu suite again (k_quads grid fix, templated angle gate, multi-block scan, wave-per-cell masks,
# hooks) + short bench + init / time-to-register timing of the BASELINE configs
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > gpurun_out/r3_run2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run2_tests.log
timeout 300 python bench.py --steps 200 --repeats 3 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register > gpurun_out/r3_run2_bench.json 2> gpurun_out/r3_run2_bench.err
echo "bench rc=$?" >> gpurun_out/r3_run2_tests.log
timeout 600 python tools/r3_init_timing.py > gpurun_out/r3_init_timing.jsonl 2> gpurun_out/r3_init_timing.err
echo "timing rc=$?" >> gpurun_out/r3_run2_tests.log
tail -30 gpurun_out/r3_run2_tests.log
