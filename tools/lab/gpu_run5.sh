#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels (default, items)"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
S4P_LIB=$R/scratch/libitems.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -3
echo "== A/B"
run() { S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab5.log; }
run super4pcs_amd/lib/libsuper4pcs_amd.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 3
run scratch/libitems.so 1
run scratch/libitems.so 3
run scratch/libitems_pipe.so 1
run scratch/libitems_pipe.so 3
echo "== PMC"
cd /tmp
for lib in scratch/libitems.so; do
  n=$(basename $lib .so)
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    S4P_LIB=$R/$lib S4P_LANES=1 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r5pmc_${n}_$i -o p --output-format csv -- python $R/tools/ab_one.py 30 1 > $R/gpurun_out/r5pmc_${n}_$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob('gpurun_out/r5pmc_*/**/p_counter_collection.csv', recursive=True)):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_verify<false>' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g'%(sum(v)/len(v)) for k,v in sorted(d.items())})
PY
