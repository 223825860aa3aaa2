#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=10
O=gpurun_out/r5_run10; mkdir -p $O
timeout -s KILL 30 python tools/r5/dbg1.py > $O/dbg.log 2>&1; echo "rc=$?" >> $O/dbg.log; tail -3 $O/dbg.log
if grep -q "perform_n_steps" $O/dbg.log; then
( timeout -s KILL 200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 150 -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
for cfg in "6 1" "9 3" "12 3" "8 2"; do set -- $cfg
  S4P_LANES=$1 S4P_GROUP=$2 S4P_TRACE_LAUNCH=1 timeout -s KILL 60 python tools/r5/tp_probe.py 300 "lanes$1_group$2" > $O/tp_$1_$2.json 2> $O/tp_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$1_$2.json")); print(d["tag"], d["runs"][0], d["best_count"], d["cand"])
except Exception as e: print("$1 $2 failed", e)
PY
  grep s4p_trace $O/tp_$1_$2.err | tail -1
done
fi
