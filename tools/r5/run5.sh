#!/bin/bash
# round 5, run 5: base groups, first light: parity tests, then throughput for group sizes / lane counts
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run5; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
( timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q --timeout 400 -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log ) 
tail -8 $O/tests.log
for cfg in "6 1" "6 3" "9 3" "12 3" "8 2" "12 2"; do set -- $cfg
  S4P_LANES=$1 S4P_GROUP=$2 S4P_TRACE_LAUNCH=1 timeout 100 python tools/r5/tp_probe.py 300 "lanes$1_group$2" > $O/tp_$1_$2.json 2> $O/tp_$1_$2.err
  python - <<PY
import json
d=json.load(open("$O/tp_$1_$2.json")); print(d["tag"], d["runs"][0], d["best_count"], d["cand"])
PY
  grep s4p_trace $O/tp_$1_$2.err | tail -1
done
