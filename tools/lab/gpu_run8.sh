#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels + registration (QLDS), kernels (no QLDS)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -4
S4P_NO_QLDS=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { S4P_LANES=$1 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab8.log; }
run 1
run 3
S4P_NO_QLDS=1 run 1
S4P_NO_QLDS=1 run 3
run 2
S4P_ABLATE=1 run 1
S4P_ABLATE=2 run 1
echo "== PMC"
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  S4P_LANES=1 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r8pmc_$i -o p --output-format csv -- python $R/tools/ab_one.py 30 1 > $R/gpurun_out/r8pmc_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob('gpurun_out/r8pmc_*/**/p_counter_collection.csv', recursive=True)):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_verify<false' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g'%(sum(v)/len(v)) for k,v in sorted(d.items())})
PY
