#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels + registration"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -4
S4P_CU_SPLIT=8 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { S4P_LANES=$1 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab9.log; }
run 1
run 3
S4P_NO_QLDS=1 run 1
S4P_CU_SPLIT=8 run 2
S4P_CU_SPLIT=8 run 3
S4P_CU_SPLIT=4 run 3
S4P_CU_SPLIT=16 run 3
S4P_CU_SPLIT=8 run 4
