#!/bin/bash
# round 4, run 10: last exploratory arms before the final pass (hardware queues, lanes), the multi-pass records test with the
# chunked FindCongruent, cold HBM points written as JSON
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run10; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_registration.py -m gpu -q -x --timeout 280 -k "multi_pass or chunked" > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3 --no-extra --no-full-count-mode"
for cfg in "S4P_X=default" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 S4P_LANES=8" "S4P_LANES=7" "GPU_MAX_HW_QUEUES=2"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
