#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=120
O=gpurun_out/r5_run25; mkdir -p $O
( timeout -s KILL 300 python -m pytest tests/test_gpu_registration.py -m gpu -x -q -k "lanes_and_base_groups" --timeout 250 -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log ); tail -5 $O/tests.log
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
