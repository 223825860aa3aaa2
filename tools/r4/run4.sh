#!/bin/bash
# round 4, run 4: batched drain in the lean sweep + per-candidate records of multi-pass bases (sink / keep / replay / sliced
# try_congruent_set) + Initialize hook: parity tests, bench arm, host-side probe, per-kernel trace with one base in flight
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py tests/test_facade.py -m gpu -q -x --timeout 400 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -12 $O/tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
for cfg in "S4P_X=default"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
timeout 120 python tools/r4/host_probe.py 300 2>&1 | tail -1 | tee $O/host_probe.json
timeout 200 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace,sq 2>&1 | tail -3
