#!/bin/bash
# round 2, pass 18: k_quads in one pass, k_pairs items per set (4096 / 8192), 768-thread verify workgroups (room for the small
# kernels next to a running verify)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels (main, v768)"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
S4P_LIB=$R/scratch/libv768.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { env S4P_LIB=$R/$1 S4P_LANES=$2 S4P_PAIR_ITEMS=$3 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | sed "s/^{/{\"pair_items\": $3, /" | tee -a gpurun_out/r2_ab18.log; }
run super4pcs_amd/lib/libsuper4pcs_amd.so 1 4096
run super4pcs_amd/lib/libsuper4pcs_amd.so 1 8192
run scratch/libv768.so 1 4096
run super4pcs_amd/lib/libsuper4pcs_amd.so 3 4096
run super4pcs_amd/lib/libsuper4pcs_amd.so 3 8192
run scratch/libv768.so 3 4096
run scratch/libv768.so 4 4096
echo "== kernel stats (1 lane, main)"
S4P_LANES=1 S4P_PREP_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2stats18 -o r -- python tools/ab_one.py 60 1 > gpurun_out/r2stats18.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2stats18/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]:
        print("  %-46s %6s %10.1f us %6s %%" % (r['Name'][:46], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
