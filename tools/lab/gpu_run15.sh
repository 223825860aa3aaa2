#!/bin/bash
# round 2, pass 15: small kernels of the base chain -- k_pairs split over (primitive, part) waves, k_prep with eight lanes per
# pair, hash table sized on the device: parity, then A/B against the previous build (scratch/libfloatlists.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels + registration"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -5
echo "== A/B"
run() { S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab15.log; }
run scratch/libfloatlists.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 3
run scratch/libfloatlists.so 3
echo "== kernel stats (1 lane)"
S4P_LANES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/r2stats15 -o r -- python tools/ab_one.py 60 1 > gpurun_out/r2stats15.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2stats15/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:44], r['Calls'], r['AverageNs'], r['Percentage'])
PY
