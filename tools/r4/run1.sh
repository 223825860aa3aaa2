#!/bin/bash
# round 4, run 1: the lean sweep (MFMA locate, coarse-only, float queries in LDS) through the kernel / registration parity
# tests, then A/B arms on one box: round-3 default (fused), staged, lean (MFMA), lean (VALU), lean at 1024 threads, lean 2 WG/CU
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run1; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 300 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -8 $O/tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
R=$GRAFT_REPO_ROOT
for cfg in "S4P_LIB=$R/scratch/libr4_fused.so S4P_NO_LEAN=1" "S4P_NO_LEAN=1" "S4P_X=lean" "S4P_LIB=$R/scratch/libr4_lean_valu.so" "S4P_VERIFY_THREADS=1024" "S4P_VERIFY_BLOCKS=512" "S4P_VERIFY_THREADS=1024 S4P_LANES=8" "S4P_LANES=4"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
tail -5 $O/err.log
