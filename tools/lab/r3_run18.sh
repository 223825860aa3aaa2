#!/bin/bash
# round 3, run 18 (the last GPU seconds of the round): the LDS-only sweep variant (scratch/libr3_staged.so) through the kernel
# parity tests, then one bench arm each
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_staged.so timeout 50 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 40 2>&1 | tail -3
for cfg in "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_staged.so" "S4P_X=0"; do
  v=$(env $cfg timeout 25 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))")
  echo "$cfg -> $v" | tee -a gpurun_out/r3_run18_ab.log
done
