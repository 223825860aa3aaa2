#!/bin/bash
# round 4, run 2: lean sweep with the VALU locate as the default -> kernel / registration parity tests; then per-kernel
# profile of a base: kernel trace + PMC passes with one base in flight, kernel trace with the default six
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run2; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 300 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 400 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 2>&1 | tail -3
timeout 120 python tools/r4/prof_kernels.py $O --steps 100 --passes trace 2>&1 | tail -3
