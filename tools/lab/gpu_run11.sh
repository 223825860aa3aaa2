#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels (dual)"
S4P_LIB=$R/scratch/libdual.so timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab11.log; }
run super4pcs_amd/lib/libsuper4pcs_amd.so 1
run scratch/libdual.so 1
run scratch/libdual.so 3
run super4pcs_amd/lib/libsuper4pcs_amd.so 3
echo "== shard tests"
timeout 900 python -m pytest tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | tail -4
