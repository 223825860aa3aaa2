#!/bin/bash
# round 3, run 3: point lists as 128-byte lines of 8 points (SoA4) + packed-FP32 exact stage: parity tests, then A/B of
# the library before (scratch/libr3_before_lines.so) and after on one box
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py tests/test_gpu_configs.py -m gpu -x -q --timeout 900 -k "not 20000" > gpurun_out/r3_run3_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run3_tests.log
B="--steps 200 --repeats 3 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register --no-parity"
for rep in 1 2; do
  S4P_LIB=$PWD/scratch/libr3_before_lines.so timeout 300 python bench.py $B > gpurun_out/r3_run3_bench_before_$rep.json 2>> gpurun_out/r3_run3_bench.err
  timeout 300 python bench.py $B > gpurun_out/r3_run3_bench_after_$rep.json 2>> gpurun_out/r3_run3_bench.err
done
S4P_VERIFY_THREADS=1024 timeout 300 python bench.py $B > gpurun_out/r3_run3_bench_after_1024.json 2>> gpurun_out/r3_run3_bench.err
python - <<'PY' >> gpurun_out/r3_run3_tests.log
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_run3_bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']/1e6,2), 'M cand/s; k_verify alone', round(d['roofline']['exclusive']['avg_launch_ms'],4), 'ms; in pipeline', round(d['roofline']['avg_launch_ms'],4))
    except Exception as e: print(f, 'ERR', e)
PY
tail -25 gpurun_out/r3_run3_tests.log
