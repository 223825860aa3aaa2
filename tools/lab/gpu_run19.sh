#!/bin/bash
# round 2, pass 19: k_verify workgroup size sweep (512 / 640 / 768 / 896 / 1024 threads, two workgroups per CU)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
run() { env S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab19.log; }
for lanes in 1 3; do
  run scratch/libv512.so $lanes
  run scratch/libv640.so $lanes
  run scratch/libv768b.so $lanes
  run scratch/libv896.so $lanes
  run super4pcs_amd/lib/libsuper4pcs_amd.so $lanes
done
