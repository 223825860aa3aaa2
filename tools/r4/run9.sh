#!/bin/bash
# round 4, run 9: lean sweep with queries from global memory (large samples): parity through the registration tests with the
# form forced at n = 2000, then the 20 000-point sample line; kernel / registration tests on the default build
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run9; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 400 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
S4P_LEAN_GLOBAL=1 timeout 400 python -m pytest tests/test_gpu_registration.py -m gpu -q -x --timeout 300 -k "early_exit or oracle or golden or whole or hippo" > $O/tests_global.log 2>&1
echo "pytest rc=$?" >> $O/tests_global.log
tail -4 $O/tests_global.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --no-full-count-mode --no-extra"
for cfg in "S4P_X=default" "S4P_NO_LEAN=1"; do
  env $cfg timeout 200 python bench.py $B --sample 20000 --steps 1 --warmup 1 --repeats 1 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']/1e6,2), 'M cand/s', round(d['ms_per_step'],1), 'ms/base', d['config']['chunked_bases'], d['provenance']['k_verify'][:120])" 2>>$O/err.log | tee -a $O/sample20000.log
done
S4P_LEAN_GLOBAL=1 timeout 100 python bench.py $B --repeats 2 2>>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lean-global at n=2000', round(d['value']/1e6,2))" | tee -a $O/sample20000.log
tail -3 $O/err.log
