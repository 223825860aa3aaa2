#!/bin/bash
# round 3, run 17: the whole GPU suite on the final default (4-chunk sweep + LDS query copy with one workgroup per CU), then --
# if minutes remain -- one short bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q -x --timeout 400 > gpurun_out/r3_run17_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run17_tests.log
tail -6 gpurun_out/r3_run17_tests.log
timeout 60 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --no-full-count-mode --repeats 3 > gpurun_out/r3_run17_bench_short.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('gpurun_out/r3_run17_bench_short.json') if l.startswith('{')][-1]); print('bench', round(d['value']/1e6,2), d['spread'])"
