#!/bin/bash
# round 4, run 3: k_pairs2 (transposed pair extraction) + candidate-record prefetch in k_verify: parity tests, A/B against the
# round-3 k_pairs, per-kernel trace with one base in flight, k_verify with the scoring ablated (fixed cost of a launch)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run3; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 300 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -6 $O/tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
for cfg in "S4P_X=default" "S4P_PAIRS_V2=0" "S4P_LANES=8" "S4P_LANES=5"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
timeout 200 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace,sq 2>&1 | tail -3
S4P_ABLATE=2 timeout 100 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace --tag ablate2 2>&1 | tail -3
S4P_ABLATE=1 timeout 100 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace --tag ablate1 2>&1 | tail -3
