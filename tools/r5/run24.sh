#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=60
O=gpurun_out/r5_run24; mkdir -p $O
timeout -s KILL 90 python tools/ab_one.py 20 3 > $O/ab20.json 2> $O/ab20.err; cat $O/ab20.json; tail -2 $O/ab20.err
timeout -s KILL 90 python tools/ab_one.py 200 2 > $O/ab200.json 2> $O/ab200.err; cat $O/ab200.json; tail -2 $O/ab200.err
