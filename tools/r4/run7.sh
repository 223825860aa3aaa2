#!/bin/bash
# round 4, run 7: heavy candidates deferred to a cooperative epilogue of k_verify: parity tests, A/B with the deferral off,
# light cycle profile, kernel trace with one base in flight
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run7; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 400 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -5 $O/tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
for cfg in "S4P_X=default" "S4P_NO_DEFER=1" "S4P_LANES=8" "S4P_LANES=4"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
B1="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 1 --no-full-count-mode"
for L in 1 6; do
S4P_LANES=$L S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr4_cycprof2.so timeout 120 python bench.py $B1 2>&1 | grep -a "cycle prof" | tail -1 | tee -a $O/cycprof2_lanes$L.log
done
timeout 100 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace 2>&1 | tail -2
S4P_NO_DEFER=1 timeout 100 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace --tag nodefer 2>&1 | tail -2
