#!/bin/bash
# round 2, pass 17: k_prep fast bucket path + k_quads one-round-trip walk + list-start alignment (1 / 4 / 8 records)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels + registration"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -5
echo "== A/B"
run() { env S4P_LIB=$R/$1 S4P_LANES=$2 S4P_LIST_ALIGN=$3 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | sed "s/^{/{\"align\": $3, /" | tee -a gpurun_out/r2_ab17.log; }
run scratch/libfloatlists.so 1 4
run super4pcs_amd/lib/libsuper4pcs_amd.so 1 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 1 4
run super4pcs_amd/lib/libsuper4pcs_amd.so 1 8
run super4pcs_amd/lib/libsuper4pcs_amd.so 3 4
run super4pcs_amd/lib/libsuper4pcs_amd.so 3 8
run scratch/libfloatlists.so 3 4
echo "== kernel stats (1 lane, new build)"
S4P_LANES=1 S4P_PREP_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2stats17 -o r -- python tools/ab_one.py 60 1 > gpurun_out/r2stats17.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2stats17/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-46s %6s %10.1f us %6s %%" % (r['Name'][:46], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
for f in glob.glob('gpurun_out/r2stats17/**/r_kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_prep' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    print("  k_prep set 1 avg %.1f us, set 2 avg %.1f us" % (sum(d[0::2]) / len(d[0::2]), sum(d[1::2]) / len(d[1::2])))
PY
