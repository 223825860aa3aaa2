#!/bin/bash
# round 4, run 19: the timed region of the bench alone on another box (the last full pass measured 163 M with a repeat at 130 M;
# the two passes before it, same four kernels, 180.6 and 181.5 M): box-to-box spread of a loop that is half host-bound
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run19; mkdir -p $O
timeout 45 python bench.py --repeats 5 --no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --no-extra --no-full-count-mode > $O/bench_short.json 2> $O/err.log
python -c "
import json; d=json.loads([l for l in open('$O/bench_short.json') if l.startswith('{')][-1]); print(round(d['value']/1e6,2), d['spread'], d['ms_per_step'], d['provenance']['git_sha'][:7])"
