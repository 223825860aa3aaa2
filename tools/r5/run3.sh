#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run3; mkdir -p $O
timeout 120 python tools/r5/verify_vs_c.py 200 > $O/verify_vs_c.json 2> $O/err.log
tail -c 6000 $O/verify_vs_c.json; tail -3 $O/err.log
