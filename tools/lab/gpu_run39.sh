#!/bin/bash
# round 2, pass 39: k_verify candidates dealt out strided over the workgroups instead of in contiguous slices
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2
run() { env S4P_LIB=$R/$1 S4P_LANES=1 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab39.log | cut -c1-330; }
run scratch/libcontig.so
run super4pcs_amd/lib/libsuper4pcs_amd.so
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'], round(d['roofline']['exclusive']['avg_launch_ms'], 4))"; }
q S4P_LIB=$R/scratch/libcontig.so
q S4P_X=1
q S4P_LIB=$R/scratch/libcontig.so
q S4P_X=1
