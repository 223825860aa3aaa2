#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab7.log; }
run super4pcs_amd/lib/libsuper4pcs_amd.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 3
run scratch/libit1.so 1
run scratch/libit2.so 1
run scratch/libit8.so 1
run scratch/libit8.so 3
echo "== PMC"
cd /tmp
S4P_LANES=1 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/r7pmc_1 -o p --output-format csv -- python $R/tools/ab_one.py 30 1 > $R/gpurun_out/r7pmc_1.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob('gpurun_out/r7pmc_*/**/p_counter_collection.csv', recursive=True)):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_verify<false>' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g'%(sum(v)/len(v)) for k,v in sorted(d.items())})
PY
